"""The reference's compression-plus-transmission energy model, restated and corrected.

Model (README.md:782-1091; tools/energy_calculator.py:28-92 in the reference): compressing S bytes costs P_c * T_c whatever
the outcome; transmitting the result costs P_t * (8 S / (CF * BW)).  Compression pays off iff
    P_c * T_c + P_t * 8S / (CF * BW)  <  P_t * 8S / BW          <=>    CF  >  E_plain / (E_plain - P_c * T_c).
The correction: the no-compression scenario spends NOTHING on compression.  The reference's CLI passes the
compression power and time into that scenario as well (its main(), lines 203-205), so it prints 851.3 Wh / 744.4 Wh / 41.4x where
its own documentation (tools/README.md:78-86) and its own break-even function give 833.3 Wh / 726.4 Wh / 40.4x; this
restatement reproduces the documented figures (tests/test_host.py).

python tools/energy_model.py --size-gb 75 --cf 9.375 --bandwidth-mbps 1 --transmit-w 5
       [--ingest-gib-s 14.5 --gpu-w 1400]   # instead of the ESP32's 0.5 W x 36 h: an MI355X at the measured ingest rate
"""
import argparse


def transmit_wh(size_gb: float, cf: float, bandwidth_mbps: float, transmit_w: float) -> float:
    seconds = size_gb * 8e9 / cf / (bandwidth_mbps * 1e6)
    return transmit_w * seconds / 3600.0


def scenario(size_gb, cf, bandwidth_mbps, transmit_w, compress_w=0.0, compress_h=0.0) -> dict:
    e_c = compress_w * compress_h
    e_t = transmit_wh(size_gb, cf, bandwidth_mbps, transmit_w)
    return {"compression_wh": e_c, "transmission_wh": e_t, "total_wh": e_c + e_t}


def breakeven_cf(size_gb, bandwidth_mbps, transmit_w, compress_w, compress_h) -> float:
    e_plain = transmit_wh(size_gb, 1.0, bandwidth_mbps, transmit_w)
    e_c = compress_w * compress_h
    return float("inf") if e_c >= e_plain else e_plain / (e_plain - e_c)


def compare(size_gb, cf, bandwidth_mbps, transmit_w, compress_w, compress_h) -> dict:
    plain = scenario(size_gb, 1.0, bandwidth_mbps, transmit_w)                      # nothing is compressed: no compression energy
    comp = scenario(size_gb, cf, bandwidth_mbps, transmit_w, compress_w, compress_h)
    saved = plain["total_wh"] - comp["total_wh"]
    return {"plain": plain, "compressed": comp, "saved_wh": saved, "saved_pct": 100.0 * saved / plain["total_wh"],
            "roi": saved / comp["compression_wh"] if comp["compression_wh"] else float("inf"),
            "breakeven_cf": breakeven_cf(size_gb, bandwidth_mbps, transmit_w, compress_w, compress_h)}


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--size-gb", type=float, default=75.0)
    ap.add_argument("--cf", type=float, default=9.375)
    ap.add_argument("--bandwidth-mbps", type=float, default=1.0)
    ap.add_argument("--transmit-w", type=float, default=5.0)
    ap.add_argument("--compress-w", type=float, default=0.5)
    ap.add_argument("--compress-h", type=float, default=36.0)
    ap.add_argument("--ingest-gib-s", type=float, default=0.0, help="derive the compression time from a measured ingest rate")
    ap.add_argument("--gpu-w", type=float, default=0.0, help="board power to use with --ingest-gib-s")
    a = ap.parse_args()
    cw, ch = a.compress_w, a.compress_h
    if a.ingest_gib_s > 0:
        ch = a.size_gb * 1e9 / (a.ingest_gib_s * 2**30) / 3600.0
        cw = a.gpu_w or cw
    r = compare(a.size_gb, a.cf, a.bandwidth_mbps, a.transmit_w, cw, ch)
    print(f"compression: {cw:g} W x {ch:.6g} h = {r['compressed']['compression_wh']:.4g} Wh")
    print(f"no compression : {r['plain']['total_wh']:.1f} Wh")
    print(f"with CF {a.cf:g}  : {r['compressed']['total_wh']:.1f} Wh  (transmission {r['compressed']['transmission_wh']:.1f} Wh)")
    print(f"saved          : {r['saved_wh']:.1f} Wh ({r['saved_pct']:.1f} %), ROI {r['roi']:.1f}x, break-even CF {r['breakeven_cf']:.4f}")


if __name__ == "__main__":
    main()

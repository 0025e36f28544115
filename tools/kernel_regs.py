"""Per-kernel register / spill table of a HIP source: python tools/kernel_regs.py hmse_amd/csrc/l1_deflate.hip [-DFLAG ...]
(compiles the device side to assembly and reads the code-object metadata; no GPU needed)."""
import re, subprocess, sys, tempfile, os
src, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out, os.path.abspath(src)] + flags,
               check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(os.path.abspath(src)) or ".")
txt = open(out).read()
md = txt[txt.index(".amdgpu_metadata"):]
for blk in md.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("dfl::", "")
    print("%-72s vgpr %3s spill %3s scratch %4s B  lds %6s" % (name[:72], g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))

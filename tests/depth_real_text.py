"""TEST INFRASTRUCTURE (uses oracle/): the oracle encoder at several chain depths against zlib level 9 on REAL text found in the image
(python sources, C headers, docs), per 8 KiB FastCDC chunk, FULL records only.  python tests/depth_real_text.py
Result (profiles/r3/r3_depth_vs_zlib9_real_text_cpu_oracle.txt): depth 32 = zlib-9 within -0.1 .. -0.6 %, depth 8 is 1-2 % LARGER — the synthetic corpora
flatter shallow depths (tools/depth_sweep.py), so level 9 keeps depth 32."""
import os, sys, zlib, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from multiprocessing import Pool
def collect(roots, exts, limit):
    out=[]; tot=0
    for root in roots:
        for dp, dn, fn in os.walk(root):
            dn.sort(); 
            for f in sorted(fn):
                if f.endswith(exts):
                    p=os.path.join(dp,f)
                    try:
                        b=open(p,'rb').read()
                    except Exception: continue
                    if len(b)<2000: continue
                    out.append(b); tot+=len(b)
                    if tot>=limit: return b''.join(out)[:limit]
    return b''.join(out)
def work(args):
    buf, depths = args
    d=np.frombuffer(buf,np.uint8)
    cfg=O.default_cfg()
    cuts=O.cdc(d,cfg)
    z=0; res={k:0 for k in depths}
    for i in range(len(cuts)-1):
        ch=buf[int(cuts[i]):int(cuts[i+1])]
        co=zlib.compressobj(9,zlib.DEFLATED,-15,9); z+=len(co.compress(ch)+co.flush())
    for k in depths:
        c=O.default_cfg(chain_depth=k)
        for i in range(len(cuts)-1):
            res[k]+=len(O.deflate(d[int(cuts[i]):int(cuts[i+1])],c))
    return z,res
if __name__=='__main__':
    sets={'py':(['/usr/lib/python3/dist-packages','/usr/local/lib/python3.10/dist-packages'],('.py',)),
          'c_headers':(['/opt/rocm/include','/usr/include'],('.h','.hpp')),
          'docs':(['/usr/share','/usr/local/lib/python3.10/dist-packages','/opt'],('.txt','.md','.rst','.html','.json','.xml','.yaml'))}
    depths=[4,8,12,16,24,32]
    for name,(roots,exts) in sets.items():
        blob=collect(roots,exts,48<<20)
        n=len(blob)//(4<<20)
        segs=[(blob[i*(4<<20):(i+1)*(4<<20)],depths) for i in range(n)]
        t=time.time()
        with Pool(8) as p: r=p.map(work,segs)
        z=sum(x[0] for x in r)
        print(name, n*4,'MiB zlib9 CF %.3f'%(n*(4<<20)/z), ' '.join('d%d:%+.2f%%'%(k,100*(sum(x[1][k] for x in r)-z)/z) for k in depths), '%.0fs'%(time.time()-t), flush=True)

"""ntt_n32768_rate.py [bits] -- standalone _NTT / _INTT at N = 32768 against batch (bits >= 53: q = 2^52 + 393217, the strict tier); HEXL_NTT_HALVES=0 for the monolithic kernels"""
import sys
sys.path[:0]=['/root/repo','/root/repo/oracle','/root/repo/tests']
import numpy as np, torch, hexl_fpga_amd as hx, orc
dev=torch.device('cuda:0'); ctx=hx.Context(0)
n=32768
bits=int(sys.argv[1]) if len(sys.argv)>1 else 51
q=orc.primes(1,bits,n)[0] if bits<53 else 4503599627763713
tb=orc.HexlTables(n,q)
tabs=[hx.as_i64(a).to(dev) for a in (tb.roots,tb.precon,tb.inv_roots,tb.inv_precon)]
base=hx.as_i64(np.stack([orc.splitmix(n,1000+b,q) for b in range(8)])).to(dev)
for batch in (256,512,2048,4096):
    x=base.repeat(batch//8,1).contiguous()
    res=[]
    for name in ("fwd","inv"):
        def run():
            if name=="fwd": ctx.ntt_fwd(x,tabs[0],tabs[1],q,n)
            else: ctx.ntt_inv(x,tabs[2],tabs[3],q,tb.inv_n,tb.inv_n_w,n)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        it=max(5, 20480//batch)
        e0.record()
        for _ in range(it): run()
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/it
        res.append(f"{name} {batch/ms*1e3/1e6:6.2f} M/s {batch*2*n*8/ms/1e6:6.0f} GB/s")
    print(f"bits {bits} batch {batch:5d}: "+"  ".join(res), flush=True)

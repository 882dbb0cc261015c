import sys
sys.path[:0]=['/root/repo','/root/repo/oracle','/root/repo/tests']
import torch, hexl_fpga_amd as hx, orc, bench
dev=torch.device('cuda:0'); ctx=hx.Context(0)
for b in (1024, 4096):
    for it in (10, 100, 1000):
        r=bench.time_ntt(hx, ctx, orc, dev, b, it)
        print(b, it, round(r['fwd']['ntt_per_s']), round(r['inv']['ntt_per_s']), round(r['fwd']['ntt_per_s_all_ranks']))

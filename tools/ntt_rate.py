import sys
sys.path[:0]=['/root/repo','/root/repo/oracle','/root/repo/tests']
import torch, hexl_fpga_amd as hx, orc, bench
dev=torch.device('cuda:0'); ctx=hx.Context(0)
for b in (1024, 4096):
    print(b, bench.time_ntt(hx, ctx, orc, dev, b, 20))

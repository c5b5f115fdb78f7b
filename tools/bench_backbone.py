"""Diagnostic: time the CNN encoder alone (B=16 images 512x768) under a few PyTorch-ROCm settings."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from unimatch_amd.encoder import CNNEncoder
def run(tag, benchmark, channels_last):
    torch.backends.cudnn.benchmark = benchmark
    net = CNNEncoder().cuda().eval()
    x = torch.randn(16, 3, 512, 768, device='cuda')
    if channels_last:
        net = net.to(memory_format=torch.channels_last); x = x.to(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3): net(x)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): net(x)
        torch.cuda.synchronize()
    print(f'{tag:40s} {(time.perf_counter()-t)/10*1e3:8.2f} ms', flush=True)
run('default', False, False)
run('benchmark=True', True, False)
run('channels_last', False, True)
run('channels_last + benchmark', True, True)

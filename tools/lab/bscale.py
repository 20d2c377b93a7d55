import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
from kbench import bench_linear
for B in (512,1024,2048,4096,512):
    bench_linear(B,48,500,1000)
for B in (1024,4096,8192):
    bench_linear(B,24,512,512)

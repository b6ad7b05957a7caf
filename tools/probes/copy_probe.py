import sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
print(os.environ.get("SOD100K_HIP_LIB", "default"), bench.measured_copy_peak(torch.device("cuda", 0)))

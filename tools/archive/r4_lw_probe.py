import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
m = bench.build_model(torch.device("cuda", 0), int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
try:
    with torch.cuda.device(0):
        m._get_handle()
    print("handle ok")
except Exception as e:
    print("ERR", e)

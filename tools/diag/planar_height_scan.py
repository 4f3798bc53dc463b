import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))  # run from the repo root
import bench_configs as bc
b = bc.b
for h in [2048, 2056, 2064, 2080, 2112, 2176, 2184, 2192, 2304, 1024, 1032, 1088, 3072]:
    plane = 480 * ((h + 7) // 8) * 128
    sys.stdout.write(f"plane={plane:#x} ")
    bc.time_blocks(f"444 planar 3840x{h}", 3840, h, b.RGB, 1, 1, 90, 1, 16, reps=50)

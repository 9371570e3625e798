"""Tests-only launcher: bench.py's launch / shard / gather logic rehearsed by gloo ranks on a machine without GPUs.

bench.py itself has no mode that runs anything but the HIP library (a benchmark must not be able to time the
checker).  This file lives under tests/, imports bench's main() and hands it the CPU oracle as the kernel module and gloo
as the process-group backend; with `--gpus N` and no WORLD_SIZE it makes bench start N ranks of THIS file.  The numbers it
prints are never reported anywhere: tests/test_distributed_cpu.py only reads the structure of the JSON line."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == '__main__':
    import bench
    from oracle import oracle
    bench.main(kernel_module=oracle, dist_backend='gloo', script=os.path.abspath(__file__))

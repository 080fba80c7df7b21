#!/usr/bin/env python3
"""Host-side cost of one eagerly launched training step (python + ctypes + torch allocator + autograd engine), which bounds
the step rate whenever the step is not replayed as a hipGraph (multi-rank runs, mini-batches).  Runs bench.py's own
main() with --no-graph under cProfile and prints the top functions by own time.  Note that autograd runs the backward
functions on its own thread: their cost shows up as time inside `run_backward`, not under their names.
    python tools/host_overhead.py [steps] [--force-dist]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == '__main__':
    steps = next((a for a in sys.argv[1:] if a.isdigit()), '200')
    sys.argv = ['bench.py', '--steps', steps, '--warmup', '10', '--no-graph', '--no-cpu-baseline', '--profile-steps', '0'] + \
        (['--force-dist'] if '--force-dist' in sys.argv else [])
    pr = cProfile.Profile()
    pr.enable()
    bench.main()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(28)

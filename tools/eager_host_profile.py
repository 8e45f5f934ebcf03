"""Where the host spends an EAGER training step (the loop the reference's own Trainer drives through the drop-in modules): cProfile over
bench.py --no-graph.  usage: python tools/eager_host_profile.py [bench args]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [os.path.join(ROOT, "bench.py"), "--no-graph", "--no-cpu-baseline", "--steps", "100"] + sys.argv[1:]
import bench

pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("cumulative").print_stats("nerf_signature_amd|torch/optim|autograd", 60)
    st.sort_stats("tottime").print_stats(25)

import sys, os, json, contextlib, time
sys.path.insert(0, os.getcwd())
import bench
with contextlib.redirect_stdout(sys.stderr):
    pass
v = bench.gp_variant(8)
print(os.path.basename(os.environ.get('GPF_LIB_PATH','default')), 'gp ms/step', round(v['ms_per_step'],3), 'variance ms', round(v['variance_pass']['ms'],3), flush=True)

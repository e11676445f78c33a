"""Throughput of the C++ pool (host/pcop_pool.*) from host memory: writes a C3 cloud as a PointCloud2 file and the C3 parameters as a
params.yaml, runs `pcop_node --gpus G --bench` (frames written in place into the pool's pinned staging, two contexts per worker) and
prints its JSON line.  usage: tools/pool_bench.py [gpus] [frames] [batch] [--same-gpu]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcloud_obstacle_processing_amd import pc2io, synth
gpus = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
same = "--same-gpu" in sys.argv
fill = "--fill" in sys.argv
HOST = os.path.join(ROOT, "pointcloud_obstacle_processing_amd", "host")
subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
p = synth.params_of("C3")
with tempfile.TemporaryDirectory() as d:
    pts = synth.make_cloud("C3", seed=3000)
    cloud = os.path.join(d, "c3.pc2")
    pc2io.write_pointcloud2(cloud, pts, point_step=16, offsets=(0, 4, 8), width=len(pts), height=1)
    yml = os.path.join(d, "params.yaml")
    base = open(os.path.join(ROOT, "tests", "golden", "params_c1.yaml")).read()
    out = []
    for ln in base.splitlines():
        key = ln.split(":")[0].strip()
        if key == "downsample_size": ln = ln.split(":")[0] + ": %g" % p["leaf"]
        elif key == "euc_cluster_tolerance": ln = ln.split(":")[0] + ": %g" % p["tol"]
        elif key == "euc_min_cluster_size": ln = ln.split(":")[0] + ": %d" % p["min_cluster"]
        elif key == "euc_max_cluster_size": ln = ln.split(":")[0] + ": %d" % p["max_cluster"]
        out.append(ln)
    open(yml, "w").write("\n".join(out) + "\n")
    args = [os.path.join(HOST, "pcop_node"), "--params", yml, "--cloud", cloud, "--gpus", str(gpus), "--bench", "--frames", str(frames), "--batch", str(batch)]
    if same:
        args.append("--same-gpu")
    if fill:
        args.append("--bench-fill")
    r = subprocess.run(args, capture_output=True, text=True, timeout=900)
    sys.stderr.write(r.stderr[-2000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode or not line:
        sys.stderr.write(r.stdout[-2000:])
        sys.exit(r.returncode or 1)
    print(json.dumps(json.loads(line[-1])))

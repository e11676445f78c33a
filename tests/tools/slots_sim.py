"""A bucket plan carried from one frame to the next, simulated on the CPU with NumPy before anything is built: per-(bucket, tile) slots sized from
frame A, frame B partitioned by A's plan - how often does a slot or a bucket overflow (= the batch falls back to the exact path)?  The plan here is a greedy packing
of the coarse bins (the library's window rule differs in detail, not in scale); bins that were empty in A belong to the bucket before them (stretched key ranges).
usage: slots_sim.py [fill=1536] [slack=2.0] [floor=16]      fill: points a planned bucket may hold (of 2048); slot capacity = slack * the bucket's fullest tile in A + floor"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pointcloud_obstacle_processing_amd import synth

fill = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
slack = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
floor_ = int(sys.argv[3]) if len(sys.argv) > 3 else 16
CAP, T, GBITS = 2048, 32768, 14
leaf = np.float32(0.02); inv = np.float32(1.0) / leaf
A = synth.make_cloud("C3", seed=3000)
others = {"another seed (boxes moved by up to 0.1 m, other sizes)": [synth.make_cloud("C3", seed=s) for s in (3001, 3002, 3003)]}
rng = np.random.default_rng(1)
stream = []
for k in range(3):   # the same scene seen again: 2 mm of noise on every point, another point order
    C = A + rng.normal(0, 0.002, A.shape).astype(np.float32)
    rng.shuffle(C, axis=0)
    stream.append(C)
others["the same scene, 2 mm noise, reshuffled"] = stream
allp = [A] + [c for v in others.values() for c in v]
lo = np.floor(np.min([c.min(0) for c in allp], 0) * inv).astype(np.int64) - 8
hi = np.floor(np.max([c.max(0) for c in allp], 0) * inv).astype(np.int64) + 8
dim = hi - lo + 1
kb = int(np.ceil(np.log2(float(dim.prod())))); cs = kb - GBITS; G = 1 << GBITS


def bins(P):
    ijk = np.floor(P * inv).astype(np.int64) - lo
    return (ijk[:, 0] + dim[0] * (ijk[:, 1] + dim[1] * ijk[:, 2])) >> cs


bA = bins(A); hA = np.bincount(bA, minlength=G)
bucket_of = np.zeros(G, dtype=np.int64); first_bin = []
nb = 0; run = 0
for b in range(G):
    if hA[b] and (nb == 0 or run + hA[b] > fill):
        nb += 1; run = 0; first_bin.append(b)
    run += hA[b]; bucket_of[b] = max(nb - 1, 0)
first_bin.append(G)
span_cells = np.diff(np.array(first_bin)) << cs          # stretched: up to the next bucket's first bin
own = np.array([(np.nonzero(hA[first_bin[i]:first_bin[i + 1]])[0].max() + 1) << cs for i in range(nb)])
nt = (len(A) + T - 1) // T


def slots(b_):
    return np.bincount(bucket_of[b_] * nt + np.arange(len(b_)) // T, minlength=nb * nt).reshape(nb, nt)


sA = slots(bA)
cap = (np.ceil(slack * sA.max(1, keepdims=True)).astype(np.int64) + floor_ + 7) // 8 * 8
print("plan of seed 3000: %d buckets at <= %d points (library today: ~590 at <= 2048), %d slots of %.0f points on average; slot capacity %.2f N records"
      % (nb, fill, sA.size, sA.mean(), float((cap * np.ones_like(sA)).sum()) / len(A)))
print("buckets the reduce can take in its direct form (span <= 2048 cells): %d of %d with their own ranges, %d stretched to the next bucket; stretched spans above 2^17 cells: %d"
      % (int((own <= 2048).sum()), nb, int((span_cells <= 2048).sum()), int((span_cells > (1 << 17)).sum())))
for name, cl in others.items():
    for C in cl:
        s = slots(bins(C))
        print("%-58s slots over capacity %3d, buckets above %d points %3d, fullest slot %.2f of its capacity, points in bins empty in the plan %.2f %%"
              % (name + ":", int((s > cap).sum()), CAP, int((s.sum(1) > CAP).sum()), float((s / cap).max()), 100.0 * float((hA[bins(C)] == 0).mean())))

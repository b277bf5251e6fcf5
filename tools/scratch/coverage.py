# dev: coverage of every genome on the reference (homologs[0][j] / L) for a bench workload
import sys, json, numpy as np, torch
sys.path.insert(0, ".")
import bench
from phylonium_amd import api
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, length, d_range, indel, inv, desc = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
buf, offs, lens = bench.make_genomes_gpu(torch, n, length, 1, dev, d_range, indel, inv, contigs=bench.CONTIGS.get(wl, 1))
ctx = api.Context(0)
ctx.set_genomes_device(buf.data_ptr(), offs, lens)

s, h = ctx.process(0)
cov = np.sort(h[0, 1:].astype(np.float64) / lens[0])
print(wl)
print("coverage deciles", np.round(cov[:: max(1, len(cov) // 20)], 4).tolist())
print("genomes with coverage < 0.001:", int((cov < 0.001).sum()), " < 0.01:", int((cov < 0.01).sum()), " < 0.1:", int((cov < 0.1).sum()), "of", len(cov))
# windows (32 positions) a genome covers at all ~ coverage for long homologies

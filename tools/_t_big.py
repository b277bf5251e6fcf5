import sys, os, time, numpy as np, torch, faulthandler, signal
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
import bench
from phylonium_amd import api
L = int(sys.argv[1]); contigs = int(sys.argv[2]); inv = float(sys.argv[3])
dev = torch.device("cuda", 0)
buf, offs, lens = bench.make_genomes_gpu(torch, 2, L, 1, dev, (0.05, 0.05), 100, inv, contigs=contigs)
torch.cuda.synchronize()
ctx = api.Context(0)
ctx.set_genomes_device(buf.data_ptr(), offs, lens)
if len(sys.argv) > 4: ctx.set_option("kmer", int(sys.argv[4]))
t = time.time(); ctx.set_reference(0); print("index", round(time.time() - t, 1), "s k", ctx.stat("index:k") if hasattr(ctx, "stat") else "", flush=True)
faulthandler.dump_traceback_later(15, exit=True)
t = time.time(); ctx.anchor(); print("anchor", round((time.time() - t) * 1e3, 1), "ms", {k: v for k, v in ctx.stats().items() if k.startswith("ms:anchor") or k.startswith("count")}, flush=True)
t = time.time(); s, h = ctx.compare(); print("compare", round((time.time() - t) * 1e3, 1), "ms", s[0, 1], h[0, 1], flush=True)

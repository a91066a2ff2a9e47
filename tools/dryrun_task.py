"""N-rank dry run of the task entry point on a 1-GPU box: builds the synthetic world of tests/test_task_gpu.py in a shared directory and runs
`emdr2_amd.tasks.run.main` on every rank.
    EMDR2_SINGLE_DEVICE=1 EMDR2_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dryrun_task.py /tmp/dry"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_task_gpu as T
tmp = sys.argv[1]
rank = int(os.environ.get("RANK", "0"))
os.makedirs(tmp, exist_ok=True)
if rank == 0:
    vocab, ev, emb = T._make_world(tmp)
    open(os.path.join(tmp, "READY"), "w").write("%s\n%s\n%s" % (vocab, ev, emb))
while not os.path.exists(os.path.join(tmp, "READY")):
    time.sleep(0.2)
vocab, ev, emb = open(os.path.join(tmp, "READY")).read().split("\n")
from emdr2_amd.tasks import run as task_run
model, results = task_run.main(T._argv(tmp, vocab, ev, emb, extra=sys.argv[2:]))          # (further task flags after the directory)
print("rank %d done: validation %s" % (rank, results.get("validation")), flush=True)

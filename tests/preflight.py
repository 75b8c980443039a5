"""Child-process legs of a `-m gpu` session, run one after the other BEFORE the pytest process touches the GPU.

tests/conftest.py starts this script at pytest_configure (a process that has initialised the GPU must not start GPU
programs on this pool; this script itself never touches the GPU -- it only starts the children) and the session waits
for it before its first GPU test, so the children never share the device with a timing- or bit-sensitive test.

  1. tests/rccl_child.py -- BASELINE config C4's collective on one GPU: a 1-rank "nccl" (= RCCL) process group, 50 MoCo
     steps through the overlapped reducer's real all-reduce Work objects (trainer/base.py:104-106);
  2. `bench.py --gpus 2 --steps 3 --warmup 1` with MF_BENCH_SHARE_GPU=1 MF_BENCH_BACKEND=gloo -- bench.py's REAL
     world > 1 worker path (self-spawned ranks, rendezvous on 127.0.0.1, the C2 main line + the C4 / C5 legs with the
     loss-partials all-reduce, MAX-over-ranks timing), both ranks on GPU 0 (the first 8-GPU driver run must not be the
     first execution of that code).

  3. `bench.py --steps 3 --warmup 1` (main line only) with MF_BENCH_FORCE_DIST=1: a 1-rank RCCL process group around the run, stdout kept
     apart -- RCCL's version banner goes through C stdio to stdout and must not land behind the compact JSON line.

usage: preflight.py <outdir>   ->   <outdir>/{rccl.log, bench2.log, bench1d.out, bench1d.log, status.json}"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, log, env, timeout):
    with open(log, "w") as fh:
        try:
            return subprocess.run(cmd, stdout=fh, stderr=subprocess.STDOUT, env=env, cwd=ROOT, timeout=timeout).returncode
        except subprocess.TimeoutExpired:
            fh.write(f"\nTIMEOUT after {timeout} s\n")
            return -9


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    status = {}
    status["rccl"] = run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py")], os.path.join(out, "rccl.log"), env, 240)
    benv = dict(env, MF_BENCH_SHARE_GPU="1", MF_BENCH_BACKEND="gloo")
    status["bench2"] = run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                           os.path.join(out, "bench2.log"), benv, 420)
    # 3. the reporting channel under a REAL RCCL communicator: bench.py with a 1-rank "nccl" group (MF_BENCH_FORCE_DIST), stdout alone
    #    in its own file -- RCCL prints its version banner through C stdio to stdout, and the compact JSON line must still be last
    denv = dict(env, MF_BENCH_FORCE_DIST="1")
    with open(os.path.join(out, "bench1d.out"), "w") as fo, open(os.path.join(out, "bench1d.log"), "w") as fe:
        try:
            status["bench1d"] = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extra-legs",
                                                "--no-cpu-baseline", "--no-train-leg"], stdout=fo, stderr=fe, env=denv, cwd=ROOT, timeout=240).returncode
        except subprocess.TimeoutExpired:
            fe.write("\nTIMEOUT after 240 s\n")
            status["bench1d"] = -9
    with open(os.path.join(out, "status.json"), "w") as fh:
        json.dump(status, fh)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Start ONE ensemble as N host processes, one per GPU -- the reference's own ensemble convention (N processes with N
ens_ids, src/greb.f90:153,1064-1068) through the drop-in boundary: every process runs

    greb_host <namelist> <proc_id> <n_procs>

in the current directory (which holds input/ and output/, as for ./greb), takes its contiguous block of the
&ENSEMBLE_PAR members (greb_host.f90; the rule of greb_climate_model_amd/ensemble.py:partition), uses GPU proc_id unless
&ENGINE_PAR device says otherwise, and writes its members' <output_file>_<ens_id> files.  The processes are started as
CHILDREN of this launcher (which never touches a GPU) -- a process that has initialised the GPU must not be re-exec'ed.
Any failing process fails the launch; the others are stopped.

  python tools/launch_ensemble.py --procs 8 [--namelist namelist] [--plan] [--serial] [--timeout S]
    --plan    every process only prints its block (no input read, no GPU touched)
    --serial  run the processes one after the other (a one-GPU box rehearsing the N-GPU run with `device = 0`)
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "greb_climate_model_amd", "greb_host")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, required=True)
    ap.add_argument("--namelist", default="namelist")
    ap.add_argument("--plan", action="store_true")
    ap.add_argument("--serial", action="store_true")
    ap.add_argument("--timeout", type=float, default=86400.0)
    ap.add_argument("--host", default=HOST)
    a = ap.parse_args(argv)
    if not os.path.exists(a.host):
        print(f"launch_ensemble: {a.host} not built (python -c 'import __graft_entry__ as g; g.build()')", file=sys.stderr)
        return 2
    cmd = lambda r: [a.host, a.namelist, str(r), str(a.procs)] + (["plan"] if a.plan else [])
    logs = [open(f"launch_ensemble.{r}.log", "w+") for r in range(a.procs)]
    rc, deadline = 0, time.monotonic() + a.timeout
    procs = []
    try:
        if a.serial:
            for r in range(a.procs):
                p = subprocess.Popen(cmd(r), stdout=logs[r], stderr=subprocess.STDOUT)
                procs.append(p)
                try:
                    code = p.wait(timeout=max(1.0, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill(); code = 124
                if code != 0:
                    rc = code if code > 0 else 1
                    print(f"launch_ensemble: process {r} exited with {code}", file=sys.stderr)
                    break
        else:
            procs = [subprocess.Popen(cmd(r), stdout=logs[r], stderr=subprocess.STDOUT) for r in range(a.procs)]
            pending = set(range(a.procs))
            while pending:
                for r in sorted(pending):
                    code = procs[r].poll()
                    if code is None:
                        continue
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code if code > 0 else 1
                        print(f"launch_ensemble: process {r} exited with {code}; stopping the others", file=sys.stderr)
                        for q in pending:
                            procs[q].terminate()
                if pending and time.monotonic() > deadline:
                    rc = rc or 124
                    for q in pending:
                        procs[q].terminate()
                    deadline = float("inf")
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, f in enumerate(logs):
        f.seek(0)
        for line in f:
            sys.stdout.write(f"[{r}] {line}")
        f.close()
        os.remove(f.name)
    return rc


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python
"""Run a command in its own process group with a time limit; at the limit the WHOLE group is killed (a plain `timeout` ends only
the first process: launchers like torch.distributed.run leave their ranks behind, spinning on the GPU under whatever runs next).

    python tools/run_group.py SECONDS cmd [args...]        -> exit code of cmd, 124 at the limit"""
import os
import signal
import subprocess
import sys


def main():
    limit = float(sys.argv[1])
    p = subprocess.Popen(sys.argv[2:], start_new_session=True)
    try:
        return p.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)                        # exactly the group started above
        p.wait()
        print("run_group: %s killed after %.0f s" % (" ".join(sys.argv[2:4]), limit), file=sys.stderr)
        return 124


if __name__ == "__main__":
    sys.exit(main())

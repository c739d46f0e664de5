#!/usr/bin/env python
"""profiles/rNN_pmc_traffic.json from the per-kernel PMC tables written by tools/rocpd_pmc.py (FETCH_SIZE and WRITE_SIZE
collected in SEPARATE rocprofv3 --pmc passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).

    python tools/pmc_to_json.py out.json fetch1.txt write1.txt [fetch2.txt write2.txt ...]
A pair written as `fetch.txt@65536 write.txt@65536` forces the units of its LANE kernels (step / act_project / gather / rollout) to
65536: the streaming rollout launches the same grid at every size, so its size is a fact of the probe, not of the grid.

traffic_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE on gfx950 reports half of the bytes of wide coalesced reads
(guide, HBM section); WRITE_SIZE is used as reported.  Keys: kernel short name -> units per launch (lanes for the env
kernels, batch rows for the update kernels), derived from the launch's total thread count."""
import json
import re
import sys

UNITS = [  # (kernel substring, threads per unit or explicit map)
    ("rollout_kernel", lambda g: g // 32),                      # 512 threads per 16 lanes
    ("rollout_stream_kernel", lambda g: 1 << 20),               # persistent: one workgroup per CU at every size (see @N above)
    ("cartsafe_step_kernel", lambda g: g if g <= 4096 else 1 << 20),
    ("cartsafe_act_project_kernel", lambda g: g if g <= 4096 else 1 << 20),
    ("replay_sample_gather_kernel", lambda g: 256 if g <= 4096 else 1 << 20),
    ("_ride_kernel", lambda g: 4096),                           # update stage at batch 256 + riders for 4096 lanes
    ("evopf_act_project_kernel", lambda g: g // 64),            # one wavefront per env lane / batch row
    ("evopf_step_kernel", lambda g: g // 64),
    ("evopf_complete_bwd_kernel", lambda g: g // 64),
]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9]+(<[A-Za-z_0-9, ]+>)?)", name)
    return m.group(1) if m else name[:60]


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+grid=(\d+)\s+n=(\d+)\s+avg=([0-9.]+)", line)
        if m:
            out[(m.group(1).strip(), int(m.group(3)))] = (m.group(2), float(m.group(5)), int(m.group(4)))
    return out


def main(out_path, files):
    workload = None
    if files and files[0] == "--workload":                      # per-workload table (profiles/rNN_pmc_traffic_<workload>.json)
        workload, files = files[1], files[2:]
    kernels = {}
    for i in range(0, len(files), 2):
        forced = None
        if "@" in files[i]:
            forced = int(files[i].rsplit("@", 1)[1])
        fetch, write = parse(files[i].rsplit("@", 1)[0] if forced else files[i]), parse(files[i + 1].rsplit("@", 1)[0] if forced else files[i + 1])
        for (name, grid), (_, fkb, n) in fetch.items():
            if "at::native" in name or "rocclr" in name or (name, grid) not in write:
                continue
            units = 256
            for sub, fn in UNITS:
                if sub in name:
                    units = fn(grid)
                    if forced and not sub.startswith("evopf") and sub != "_ride_kernel":
                        units = forced
            wkb = write[(name, grid)][1]
            if str(units) in kernels.get(short(name), {}):      # earlier files win (cart-DDPG iterations before the cart-SAC ride probe)
                continue
            kernels.setdefault(short(name), {})[str(units)] = {
                "grid_threads": grid, "launches": n, "fetch_kb_raw": round(fkb, 1), "write_kb": round(wkb, 1),
                "traffic_bytes": int(round((2 * fkb + wkb) * 1024))}
    note = ("HBM traffic per launch from rocprofv3 PMC counters (separate --pmc passes: FETCH_SIZE, WRITE_SIZE; units KB). "
            "Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports 1/2 of the bytes of wide "
            "coalesced reads and is doubled; WRITE_SIZE is used as reported. Recipe: tools/collect_profiles.sh "
            "(tools/kernel_probe.py step | iter + tools/rocpd_pmc.py). The update kernels' traffic at batch 256 is dominated "
            "by each XCD's L2 fetching its own copy of the weights it touches.")
    doc = {"_note": note, "kernels": kernels}
    if workload:
        doc["workload"] = workload
        doc["_note"] += " This table: tools/kernel_probe.py window:%s (policy_fre periods of that bench trainer, eagerly)." % workload
    json.dump(doc, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])

#!/usr/bin/env python3
"""Generate the committed data fixtures from the reference checkout (run in the build container only).

Reads DATA from /root/reference (URDF robot descriptions, a documentation table, one trajectory
npz) and writes small derived fixtures; no reference source code is copied.

  flobaroid_amd/robots/<name>.topology.json   kinematic tree + a-priori parameters extracted from
                                              model/<name>.urdf by flobaroid_amd.topology.parse_urdf
  tests/golden/kuka_tutorial_apriori.json     xStdModel[0:101] (8 decimals) and the nID flags printed in
                                              documentation/TUTORIAL.md:60-160  (known answer F9)
  tests/golden/kuka_trajectory_opt_1.npz      2409x7 positions/velocities/accelerations + times of
                                              model/kuka_lwr4.urdf.trajectory_opt_1.npz and its recorded
                                              n_observable_base_params (= 64)            (known answer F5)
  tests/golden/structure.json                 documented structure counts (links/DOF/base ranks)
"""
import json
import os
import re
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"

from flobaroid_amd.topology import parse_urdf  # noqa: E402


def main():
    robots = os.path.join(REPO, "flobaroid_amd", "robots")
    golden = os.path.join(REPO, "tests", "golden")
    os.makedirs(robots, exist_ok=True)
    os.makedirs(golden, exist_ok=True)
    for name in ["threeLinks", "kuka_lwr4", "walkman_left_arm", "walkman_apriori"]:
        t = parse_urdf(os.path.join(REF, "model", name + ".urdf"))
        t.save_json(os.path.join(robots, name + ".topology.json"))
        print(name, t.num_links, "links", t.num_dofs, "dofs")

    # --- F9: TUTORIAL table -------------------------------------------------------------
    rows = []
    pat = re.compile(r"^\|\s*(-?\d+\.\d+)\|\s*(-?\d+\.\d+)\|[^|]*\|[^|]*\|([^|]*)\|#(\d+): (\S+) - (.*)$")
    with open(os.path.join(REF, "documentation", "TUTORIAL.md")) as f:
        for line in f:
            m = pat.match(line.rstrip("\n"))
            if m:
                rows.append((int(m.group(4)), float(m.group(1)), "nID" in m.group(3), m.group(5), m.group(6)))
    rows.sort()
    assert [r[0] for r in rows] == list(range(101)), len(rows)
    with open(os.path.join(golden, "kuka_tutorial_apriori.json"), "w") as f:
        json.dump(
            {
                "source": "documentation/TUTORIAL.md:60-160 (A priori column, nID flag)",
                "xStdModel": [r[1] for r in rows],
                "non_id": [r[0] for r in rows if r[2]],
                "symbols": [r[3] for r in rows],
                "descriptions": [r[4] for r in rows],
                "apriori_mass": 16.0,
            },
            f,
            indent=0,
        )

    # --- F5: trajectory fixture ---------------------------------------------------------
    z = np.load(os.path.join(REF, "model", "kuka_lwr4.urdf.trajectory_opt_1.npz"), allow_pickle=True)
    np.savez_compressed(
        os.path.join(golden, "kuka_trajectory_opt_1.npz"),
        positions=z["positions"],
        velocities=z["velocities"],
        accelerations=z["accelerations"],
        times=z["times"],
        frequency=z["frequency"],
        n_observable_base_params=z["n_observable_base_params"],
    )

    # --- documented structure counts ----------------------------------------------------
    with open(os.path.join(golden, "structure.json"), "w") as f:
        json.dump(
            {
                "source": "documentation/analysis_findings.md:29,43,45; design_notes.md:98-104; "
                "model/kuka_lwr4.urdf.trajectory_opt_1.npz; SURVEY.md Appendix A probe table",
                "threeLinks": {"links": 3, "dofs": 2, "base_rank_floating": 24},
                "kuka_lwr4": {"links": 8, "dofs": 7, "base_rank_fixed": 43, "base_rank_fixed_friction": 64,
                              "mass": 16.0},
                "walkman_left_arm": {"links": 9, "dofs": 7, "base_rank_floating": 59},
                "walkman_apriori": {"links": 48, "dofs": 29, "base_rank_floating": 213, "mass": 128.0,
                                    "free_links": 30},
            },
            f,
            indent=1,
        )


if __name__ == "__main__":
    main()

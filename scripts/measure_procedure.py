#!/usr/bin/env python3
"""The reference's own benchmark procedure, recorded (round-3 verdict item 7).

src/bin/measure.rs:27-154: load <scene.ply> + <cameras.json>, a 2048x2048 Rgba8Unorm target, every TRAIN camera (7 of 8,
scene.rs:143-151) ten times, frames recorded back to back on one queue with ONE wait at the end, the clock started before
the warm-up frame; prints "average FPS".  README.md:55 publishes "> 200 FPS" for the real bonsai scene at 1200x799 on a
3090 -- the asset is not in this environment, so the scene here is the seeded bonsai-like stand-in (c2: 1.2 M Gaussians),
written as a real INRIA-layout .ply and a real cameras.json and read back by the library's loaders.

    python scripts/measure_procedure.py [out.json] [n_gaussians] [n_cameras]

Runs the `websplat_measure` binary (tools/measure.cpp: a thin main over ws_measure) exactly as one would run the
reference's `measure`, once with one frame in flight (the reference's procedure) and once with four.
"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd")]
from websplat import synth  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "measure_rs_procedure.json")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_200_000
    ncam = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    binary = os.path.join(ROOT, "web-splat_amd", "bin", "websplat_measure")
    if not os.path.exists(binary):
        subprocess.run(["make", "-C", os.path.join(ROOT, "web-splat_amd"), "all"], check=True, capture_output=True)
    res = {"procedure": "src/bin/measure.rs:27-154 via tools/measure.cpp -> ws_measure: 2048x2048, Rgba8Unorm, 10 x every train "
                        "camera, back to back, one sync, clock started before the warm-up frame",
           "scene": f"seeded bonsai-like stand-in (synth.scene_c2, {n} Gaussians, sh_deg 3) written as .ply + cameras.json",
           "cameras": ncam, "train_cameras": sum(1 for i in range(ncam) if i % 8 != 0),
           "reference_published": "README.md:55: > 200 FPS, bonsai at 1200x799 on an RTX 3090 (other hardware, other scene file)"}
    with tempfile.TemporaryDirectory() as td:
        ply, cj = os.path.join(td, "point_cloud.ply"), os.path.join(td, "cameras.json")
        synth.write_ply(ply, synth.scene_c2(n=n, seed=1), 3)
        synth.write_cameras_json(cj, synth.orbit_cameras(ncam, 1559, 1039, 1160.0, 1160.0))
        env = dict(os.environ)
        env.setdefault("GPU_MAX_HW_QUEUES", "8")
        for inflight in (1, 4):
            runs = []
            for rep in range(3):
                t0 = time.time()
                p = subprocess.run([binary, ply, cj, str(inflight)], capture_output=True, text=True, env=env, timeout=900)
                m = re.search(r"average FPS: ([0-9.eE+-]+)", p.stdout)
                if p.returncode != 0 or not m:
                    raise SystemExit(f"websplat_measure failed ({p.returncode}): {p.stdout[-500:]} {p.stderr[-500:]}")
                runs.append({"average_fps": float(m.group(1)), "wall_s": round(time.time() - t0, 2)})
            res[f"frames_in_flight_{inflight}"] = {"runs": runs, "average_fps_best": max(r["average_fps"] for r in runs),
                                                   "average_fps_median": sorted(r["average_fps"] for r in runs)[1]}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

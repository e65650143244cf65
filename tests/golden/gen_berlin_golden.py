"""Flattens the only Ceres-produced artefact the reference holds -- data/berlin/reconstruction_example.json (a converged OpenSfM
reconstruction: 3 shots, 1430 points, one perspective camera with free k1 / k2 / focal) with the observations of
data/berlin/tracks_example.csv and the control points of ground_control_points.json -- into tests/golden/berlin_example.json, so that
the tests can load it where /root/reference is not mounted (the GPU box).  Run from the repo root: python tests/golden/gen_berlin_golden.py"""
import json
import os

REF = "/root/reference/data/berlin"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "berlin_example.json")


def main():
    r = json.load(open(os.path.join(REF, "reconstruction_example.json")))[0]
    out = {"source": "mapillary/OpenSfM data/berlin: reconstruction_example.json + tracks_example.csv + ground_control_points.json",
           "reference_lla": r["reference_lla"], "cameras": r["cameras"], "rig_cameras": r["rig_cameras"], "rig_instances": r["rig_instances"],
           "shots": {k: {f: v[f] for f in ("rotation", "translation", "camera", "gps_position", "gps_dop")} for k, v in r["shots"].items()},
           "points": {k: v["coordinates"] for k, v in r["points"].items()}, "observations": [], "gcp": []}
    for line in open(os.path.join(REF, "tracks_example.csv")).read().splitlines()[1:]:
        f = line.split("\t")
        if f[1] in r["points"] and f[0] in r["shots"]:
            out["observations"].append([f[0], f[1], float(f[3]), float(f[4]), float(f[5])])  # image, track, x, y, scale (tracking.py:108)
    out["gcp"] = json.load(open(os.path.join(REF, "ground_control_points.json")))["points"]
    json.dump(out, open(OUT, "w"), separators=(",", ":"))
    print(OUT, os.path.getsize(OUT), "bytes;", len(out["points"]), "points,", len(out["observations"]), "observations")


if __name__ == "__main__":
    main()

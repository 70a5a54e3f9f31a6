"""Generates tests/golden/bspline_notebook.json by EXECUTING the reference's own numpy prototype
(/root/reference/scripts/CubicBSpline3D.ipynb, same basis matrix and sample data as
src/odometry/spline_interpolation_test.cc:79-96) in this container.  Only the resulting numbers are
committed; the notebook text is read from the read-only reference checkout at generation time and is not
stored here.  Run:  python tests/golden/make_bspline_golden.py
"""
import json
import os
import sys
import types

NB = "/root/reference/scripts/CubicBSpline3D.ipynb"


def main():
    nb = json.load(open(NB))
    src = "".join(nb["cells"][0]["source"])
    src = src.split("# Spline display")[0]  # drop the plotting tail
    sys.modules.setdefault("matplotlib", types.ModuleType("matplotlib"))
    sys.modules.setdefault("matplotlib.pyplot", types.ModuleType("matplotlib.pyplot"))
    ns = {}
    exec(compile(src, NB, "exec"), ns)  # runs the reference prototype
    Nq, Nbs = ns["Nq"], ns["Nbs"]
    f = [Nq * (i / Nbs) for i in range(1, Nbs + 1) if Nq * (i / Nbs) >= 1]
    out = {
        "source": "scripts/CubicBSpline3D.ipynb executed in-container",
        "p": ns["p"].tolist(),
        "Q": ns["Q"].tolist(),
        "index_f": f,
        "curve": ns["BSpline"].tolist(),
    }
    assert len(out["curve"]) == len(f)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bspline_notebook.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, len(f), "samples")


if __name__ == "__main__":
    main()

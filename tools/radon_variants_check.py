"""The four forward-projector kernels (quads of symmetric angles, window-sharing, per-wave LDS windows, direct gathers) must agree to rounding on
random geometries: run each in its own process (the choice is read once per process) and compare."""
import sys, os, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from trips_py_amd.operators import Radon2DParallel
    rng = np.random.default_rng(123)
    out = []
    for N, na, nd in ((1024, 9, 1024), (1100, 13, 900), (2048, 6, 2500), (1536, 17, 1536)):
        ang = np.sort(rng.uniform(0, np.pi, na)) if N != 1100 else rng.uniform(-3, 6, na)
        R = Radon2DParallel(N, ang, n_det=nd)
        x = torch.rand(N * N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(N))
        y = R.apply(x)
        out.append(y.double().cpu().numpy().tolist())
    json.dump(out, open(sys.argv[2], "w"))
else:
    import numpy as np
    res = {}
    for name, env in (("quad", {}), ("win", {"TRK_RADON_NO_QUAD": "1"}), ("lds", {"TRK_RADON_NO_WIN": "1"}), ("direct", {"TRK_RADON_NO_LDS": "1"})):
        f = f"/tmp/radon_{name}.json"
        r = subprocess.run([sys.executable, __file__, "child", f], env=dict(os.environ, **env), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-800:]
        res[name] = [np.array(v) for v in json.load(open(f))]
    for i in range(len(res["win"])):
        assert len(res["quad"]) == len(res["win"])
        ref = res["direct"][i]
        for name in ("quad", "win", "lds"):
            e = np.abs(res[name][i] - ref).max() / np.abs(ref).max()
            print(f"case {i}: {name} vs direct max rel diff {e:.2e}")
            assert e < 2e-6
    print("OK")

import importlib, os, sys, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
syn = importlib.import_module("ai-camera_amd.synthetic")
ef = importlib.import_module("ai-camera_amd.engine_file")
TP = importlib.import_module("ai-camera_amd.pipeline").TrackingPipeline
y, r = ef.ensure_seeded_engines(ROOT)
n_frames, batch = 40, 8
births = {t: 4 + (t - 40) // 3 for t in range(40, 76)}
sc = syn.Scene(seed=31, n_targets=76, births=births, w_range=(30.0, 50.0), h_range=(90.0, 140.0))
frames = sc.render_batch(0, n_frames)
res = {}
for mode in (0, 2, 1):
    pipe = TP(y, r, (720, 1280), batch=batch, ring_frames=n_frames, max_persons=80, dtype="fp16", inject=True)
    pipe.option("device_assoc", mode)
    pipe.upload(0, frames)
    pipe.inject(0, [sc.detections(f)[:3] for f in range(n_frames)])
    nt, rows, nd = pipe.run_raw(0, n_frames)
    res[mode] = (nt.copy(), rows.copy())
    pipe.close()
for m in (2, 1):
    for f in range(n_frames):
        a, b = res[0][1][f][:res[0][0][f]], res[m][1][f][:res[m][0][f]]
        if a.shape != b.shape or not np.array_equal(a, b):
            d = np.abs(a - b).max(1) if a.shape == b.shape else None
            print("mode", m, "first diff at frame", f, "shapes", a.shape, b.shape, "rows differing", None if d is None else np.nonzero(d)[0].tolist(), None if d is None else a[d > 0][:3], None if d is None else b[d > 0][:3])
            break
    else:
        print("mode", m, "identical to host mode on all frames")

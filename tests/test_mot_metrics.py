"""MOT quality harness (ai-camera_amd/mot_metrics.py): known-answer sequences on the host, and the pipeline's tracks on a planted
scene with occlusion gaps and late births scored against the scene's own identities."""
import numpy as np
import pytest

from conftest import pkg

mm = pkg("mot_metrics")


def _box(x):
    return [x, 10.0, x + 20.0, 60.0]


def test_known_answers():
    # two targets, 10 frames; the tracker follows both but swaps their ids from frame 6 on, misses target 1 in frame 3 and
    # hallucinates one box in frame 8
    gt, out = [], []
    for f in range(10):
        gt.append((np.array([_box(10 + f), _box(200 - f)]), [0, 1]))
        ids = (7, 9) if f < 6 else (9, 7)
        o = [(*_box(10 + f), ids[0])]
        if f != 3:
            o.append((*_box(200 - f), ids[1]))
        if f == 8:
            o.append((500.0, 500.0, 520.0, 560.0, 42))
        out.append(o)
    r = mm.evaluate(gt, out)
    assert (r["gt"], r["outputs"], r["fn"], r["fp"], r["idsw"]) == (20, 20, 1, 1, 2)
    assert r["mota"] == pytest.approx(1 - 4 / 20)
    # identity measures: best global assignment keeps 0->7 (6 frames) and 1->9 (5 frames: frame 3 is missed)  => IDTP 11
    assert r["idf1"] == pytest.approx(2 * 11 / 40) and r["idp"] == pytest.approx(11 / 20) and r["idr"] == pytest.approx(11 / 20)
    perfect = mm.evaluate(gt, [[(*b, i + 1) for b, i in zip(g[0].tolist(), g[1])] for g in gt])
    assert perfect["mota"] == 1.0 and perfect["idf1"] == 1.0 and perfect["idsw"] == 0
    empty = mm.evaluate(gt, [[] for _ in gt])
    assert empty["mota"] == 0.0 and empty["fn"] == 20 and empty["idf1"] == 0.0
    assert mm.iou_matrix([[0, 0, 10, 10]], [[5, 5, 15, 15]])[0, 0] == pytest.approx(25 / 175)


@pytest.mark.gpu
def test_pipeline_quality_on_planted_scene(gpu, engines):
    """The tracker's own quality on a scene whose identities are known: 16 persons, two occlusion gaps, one late birth, 80 frames.
    Only the confirmation delay (n_init = 3 frames per new track) and the gaps may cost recall; no identity may switch."""
    syn = pkg("synthetic")
    n_frames = 80
    sc = syn.Scene(seed=17, n_targets=16, gaps=[(3, 20, 30), (9, 40, 44)], births={12: 25})
    frames = sc.render_batch(0, n_frames)
    TP = pkg("pipeline").TrackingPipeline
    pipe = TP(engines[0], engines[1], (720, 1280), batch=16, ring_frames=n_frames, max_persons=32, dtype="fp16", inject=True)
    pipe.upload(0, frames)
    pipe.inject(0, [sc.detections(f)[:3] for f in range(n_frames)])
    tracks, _ = pipe.run(0, n_frames)
    r = mm.evaluate(mm.scene_ground_truth(sc, n_frames), tracks)
    print(f"planted scene: MOTA {r['mota']:.4f} IDF1 {r['idf1']:.4f} FP {r['fp']} FN {r['fn']} IDSW {r['idsw']} of {r['gt']} boxes")
    assert r["idsw"] == 0 and r["fp"] == 0
    assert r["fn"] <= 2 * 17 + 4 and r["mota"] > 0.95 and r["idf1"] > 0.97       # 2 unconfirmed frames per track birth (16 + 1 late)
    pipe.close()

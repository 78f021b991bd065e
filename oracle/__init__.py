"""CPU oracle for the per-frame detect+track hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product path
(``ai-camera_amd/`` + ``libaicam.so``) never imports, links or executes it and
fails loudly when the HIP library is missing.

Parity status:
  * ``deepsort_oracle``  -- PINNED: checked against the reference's own
    ``src/tracker/core`` (imported in the build container only) through the
    committed fixtures in ``tests/golden/`` and the known-answer constants of
    the reference's ``__main__`` self-tests.
  * ``lsap_oracle``      -- PINNED against ``scipy.optimize.linear_sum_assignment``
    (scipy 1.15.3 == the reference's pin, requirements.txt:8).
  * ``image_oracle``     -- formulas follow ``src/utils/image_processing.py`` as
    text; the u8 resize arithmetic restates OpenCV 4.11 (opencv-python==4.11.0.86,
    requirements.txt:3), which is absent here: **parity unpinned** for the
    bit pattern of cv2.resize, the contract is the restated fixed-point spec.
  * ``nets_oracle``      -- the conv arithmetic of the reference lives in
    TensorRT engines built from un-vendored ONNX files: **parity unpinned**;
    the oracle is a plain PyTorch-CPU fp32 interpreter of the same engine file.
"""

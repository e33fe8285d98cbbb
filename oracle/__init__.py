"""oracle/ — CPU restatement of the reference algorithm for the DRecPy hot path.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import anything from this package, and only as the checker.  The product
(`drecpy_amd/`) never imports it and fails loudly when its HIP library is missing.

Parity status (also stated in DESIGN.md):
  * id map, PointSampler / ListSampler streams, interaction vectors, corruption-mask stream,
    ranking metrics: PINNED against the reference itself — `oracle/gen_golden.py` imports the
    reference's Dataset/Sampler/Evaluation code in the dev container and the resulting vectors are
    committed under tests/golden/ (the reference's own known-answer tests are included, e.g.
    tests/Dataset/test_mem_dataset.py:678-697).
  * CDAE arithmetic (forward, Keras BCE/MSE with the (B,B,N) broadcast, L2, Keras Adam with the
    per-variable step counter): **parity unpinned** — TensorFlow is a third-party dependency
    (requirements.txt:6, `tensorflow>=2.0<3`, un-pinned) that is absent from this image and the
    reference holds no test or golden vector for any deep model.  The restatement follows
    DRecPy/Recommender/cdae.py and recommender_abc.py line by line plus the published TF-2.x /
    Keras numerics, and is cross-checked against torch-CPU autograd.
"""

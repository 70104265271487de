"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's per-pixel video post-processing path
(Fast Film Grain, 3D-LUT apply, Color Match, Unsharp/Laplacian/Sobel sharpen).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and there only as the *checker* -- never as the
thing measured or shipped.  The product package (``comfyui-vrgamedevgirl_amd``)
never imports it and has no CPU fallback: it fails loudly when the HIP
extension is missing.

Pinning status (see DESIGN.md "Oracle"):
  * grain / LUT / unsharp / laplacian / sobel: PINNED -- ``oracle/make_golden.py``
    imports the reference's own ``nodes.py`` / ``VRGDG_IV_Adjustments.py`` /
    ``VRGDG_LUTVideoTools.py`` / ``VRGDG_StandaloneVideoEnhancerNodes.py`` from
    ``/root/reference`` (with ``sys.modules`` stubs for ComfyUI) and commits the
    outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks the
    restatement against them bit-for-bit.
  * colour match: PARITY UNPINNED for the Lab transforms -- the arithmetic
    lives in third-party ``kornia.color.rgb_to_lab / lab_to_rgb`` (unpinned in
    the reference's requirements.txt:1, not installed here, not installable).
    The statistics / blend logic (nodes.py:91-124) *is* pinned by running the
    reference's ``match_color`` with the restated Lab functions injected as the
    ``kornia`` stub.
"""

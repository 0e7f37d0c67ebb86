"""The product's host-side cull (csrc/svr_cull.h: is_visible with four matrix rows per operation) against the oracle's
scalar restatement of src/vk_engine.cpp:56-86 on two million random objects — dense and projective matrices, w crossing
zero, zero, infinite and NaN boxes.  Every verdict must agree: culled draws change what is drawn.  CPU only."""
import os
import subprocess

import __graft_entry__ as g


def test_vector_cull_agrees_with_the_oracle(tmp_path):
    g.build_oracle() if hasattr(g, "build_oracle") else None
    exe = str(tmp_path / "cull_check")
    src = os.path.join(g.ROOT, "tests", "native", "cull_check.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-o", exe, src, "-ldl"], check=True)
    ora = os.path.join(g.ROOT, "oracle", "libsvr_oracle.so")
    assert os.path.exists(ora), "oracle library not built"
    for seed in (1, 2):
        r = subprocess.run([exe, ora, "1000000", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        words = r.stdout.split()
        assert words[0] == "mismatches" and words[1] == "0", r.stdout
        assert 50000 < int(words[-1]) < 950000, "the sample should hold both verdicts: " + r.stdout

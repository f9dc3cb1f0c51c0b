"""Static properties of the compiled GPU kernels (no GPU needed): nothing in the library spills to private
memory or reads the dispatch packet -- either turns a kernel several times slower without changing a result,
so no parity test would notice."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_kernel_uses_scratch_or_the_dispatch_packet():
    import check_code_objects

    lines, problems = check_code_objects.check(os.path.join(ROOT, "detr_amd", "libdetr_match.so"))
    assert len(lines) >= 40, "expected the cost, assignment, pack, consumer and auxiliary kernels"
    assert not problems, "\n".join(problems)

"""detr_amd -- the DETR-family bipartite-matching path on MI355X (gfx950).

``HungarianMatcher`` is a drop-in for the reference's ``models.matcher.HungarianMatcher``;
the work is done by hand-written HIP kernels in ``libdetr_match.so`` (C ABI declared in
``include/detr_match.h``): the matching cost and the batched assignment, and -- for the
criterion's steps right behind the matcher (``losses``, ``aux_assign``, ``criterion.attach``) --
the losses of the matched pairs, the class maps, the classification loss and Co-DETR's auxiliary
assigners.  Importing the package fails if that library has not been built.
"""
from ._lib import LIB_PATH, lib  # noqa: F401  (fails loudly when the extension is missing)
from .matcher import FlatMatches, HungarianMatcher, PackedTargets  # noqa: F401
from .lsap import batched_linear_sum_assignment  # noqa: F401
from . import aux_assign, criterion, dist, losses, synth  # noqa: F401

__all__ = ["HungarianMatcher", "PackedTargets", "FlatMatches", "batched_linear_sum_assignment", "aux_assign", "criterion", "dist", "losses", "synth", "LIB_PATH"]

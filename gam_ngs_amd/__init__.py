"""gam_ngs_amd -- MI355X (gfx950) implementation of gam-merge's contig-pair alignment hot path.

The product is the C-ABI shared library ``libgamdp.so`` (see ``include/gamdp.h``), built from
``gam_ngs_amd/csrc``.  This package is only its Python face: a ctypes binding (``gam_ngs_amd.lib``)
and a host-side mirror of the reference's operator interface for the path (``gam_ngs_amd.api``:
``BandedSmithWaterman``, ``MyAlignment``, ``ABlast``, ``PctgBuilder.alignMergeBlock``), used by the
tests and the benchmark.  There is no CPU fallback: importing works anywhere (the library loads
without a GPU so its symbols can be checked), but creating a context without a gfx950 device raises.
"""
from .lib import GamdpError, load_library, library_path  # noqa: F401
from .api import (ABlast, BandedSmithWaterman, Block, Context, MergeBlock, MultiContext, MultiSequenceSet,  # noqa: F401
                  MyAlignment, PctgBuilder, SequenceSet, GAP_A, GAP_B, MATCH, MISMATCH)

__all__ = ["GamdpError", "load_library", "library_path", "Context", "SequenceSet", "BandedSmithWaterman",
           "MyAlignment", "ABlast", "MultiContext", "MultiSequenceSet", "PctgBuilder", "MergeBlock", "Block", "GAP_A", "GAP_B", "MATCH", "MISMATCH"]

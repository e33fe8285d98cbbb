"""Import recipe for the reference (dev container only).

TEST INFRASTRUCTURE.  Only oracle/gen_golden.py uses this, to produce the committed fixtures under
tests/golden/.  /root/reference does not exist on the GPU box; nothing at test/bench time imports it.
"""
import os
import sys
import tempfile

REFERENCE_ROOT = '/root/reference'


def import_reference():
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError('reference tree not present (this only runs in the dev container)')
    sys.dont_write_bytecode = True              # never drop __pycache__ into the read-only tree
    os.environ.setdefault('DATA_FOLDER', tempfile.mkdtemp(prefix='drecpy_data_'))
    os.environ.setdefault('MPLBACKEND', 'Agg')
    import numpy as np
    if not hasattr(np, 'float'):
        np.float = float                        # mem_dataset.py:150 uses the removed alias
    here = os.path.dirname(os.path.abspath(__file__))
    stubs = os.path.join(here, '_stubs')
    for p in (REFERENCE_ROOT, stubs):
        if p not in sys.path:
            sys.path.insert(0, p)
    import DRecPy  # noqa: F401
    return DRecPy

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'bm-nas_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter):
    """Which evaluation of the reference math every whole-step comparison matched (gpu_util.match_step): visible
    under -q too, so that a drift from "fp32" towards rescued matches cannot go unnoticed."""
    try:
        import gpu_util
    except Exception:                                    # noqa: BLE001 — CPU-only sessions may never import it
        return
    out = getattr(gpu_util, 'OUTCOMES', [])
    if not out:
        return
    kinds = {}
    for _, how in out:
        k = how if how in ('fp32', 'fp64') else 'fp64+flips'
        kinds[k] = kinds.get(k, 0) + 1
    tr = terminalreporter
    tr.write_line('match_step: ' + ', '.join(f'{v} x {k}' for k, v in sorted(kinds.items())) +
                  f' of {len(out)} whole-step comparisons')
    for label, how in out:
        if how != 'fp32':
            tr.write_line(f'  match_step [{label}] -> {how}')

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def engine():
    import smoothxg_amd as S
    eng = S.PoaEngine(0)
    yield eng
    eng.close()


@pytest.fixture(autouse=True, scope="session")
def gfa_invariants_on_every_output():
    """Every GFA the suites get out of the host library -- whole iterations (with and without the MAF consumer) and single block
    graphs, on CPU with the oracle-backed provider and on the GPU -- is also run through tests/gfa_invariants.py: checks that
    share no code with the product or with oracle/smooth_oracle.py (edges == walked step pairs, unchopped, topological block
    graphs, input paths preserved).  SXG_TEST_NO_INVARIANTS=1 switches it off."""
    if os.environ.get("SXG_TEST_NO_INVARIANTS"):
        yield
        return
    try:
        from smoothxg_amd import smooth as SM
    except Exception:
        yield
        return
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gfa_invariants as GI
    orig_init, orig_gfa, orig_maf, orig_bg = SM.Smoother.__init__, SM.Smoother.smooth_gfa, SM.Smoother.smooth_maf_gfa, SM.Smoother.block_graph_gfa
    checked = {"laced": 0, "block": 0}

    def init(self, gfa_text, *a, **k):
        self._input_text = gfa_text.decode() if isinstance(gfa_text, bytes) else gfa_text
        orig_init(self, gfa_text, *a, **k)

    def smooth_gfa(self, params, provider):
        out = orig_gfa(self, params, provider)
        if out is not None:
            GI.check_laced(out, self._input_text)
            checked["laced"] += 1
        return out

    def smooth_maf_gfa(self, params, provider, *a, **k):
        res = orig_maf(self, params, provider, *a, **k)
        GI.check_laced(res[0], self._input_text)
        checked["laced"] += 1
        return res

    def block_graph_gfa(self, block_id, params, provider):
        out = orig_bg(self, block_id, params, provider)
        GI.check_block_graph(out)
        checked["block"] += 1
        return out

    SM.Smoother.__init__, SM.Smoother.smooth_gfa, SM.Smoother.smooth_maf_gfa, SM.Smoother.block_graph_gfa = init, smooth_gfa, smooth_maf_gfa, block_graph_gfa
    yield checked
    SM.Smoother.__init__, SM.Smoother.smooth_gfa, SM.Smoother.smooth_maf_gfa, SM.Smoother.block_graph_gfa = orig_init, orig_gfa, orig_maf, orig_bg

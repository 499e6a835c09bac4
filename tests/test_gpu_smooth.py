"""GPU: one smoothing iteration end to end -- collection (A2-A4) -> ONE batched GPU POA call ->
block graphs (A9/A10) -> lacing -> GFA -- through the two C ABIs, against the oracle stack
(oracle/smooth_oracle.py + oracle/poa_oracle.c) on the reference's own DRB1 test input."""
import os

import pytest

from oracle import smooth_oracle as SO
from smoothxg_amd import smooth as S

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DRB1 = os.path.join(HERE, "golden", "DRB1-3123.seqwish.gfa")


@pytest.mark.parametrize("cons", [0, 1])
def test_drb1_smoothing_iteration_on_gpu(engine, cons):
    text = open(DRB1).read()
    g = SO.Graph(text)
    sm = S.Smoother(text, 700)           # the reference's ctest uses -l 700,900,1100 (CMakeLists.txt:565)
    got = sm.smooth_gfa(S.default_params(add_consensus=cons), S.gpu_provider(engine))
    want = SO.smooth(g, SO.blockset_by_path_windows(g, 700), add_consensus=bool(cons))
    assert got == want
    out = SO.Graph(got)
    for q, nm in enumerate(g.pname):     # src/main.cpp:770-803
        assert out.path_sequence(out.pname.index(nm)) == g.path_sequence(q)
    assert len(out.pname) == 12 + (20 if cons else 0)


def test_global_alignment_and_block_graph_on_gpu(engine):
    text = open(DRB1).read()
    g = SO.Graph(text)
    sm = S.Smoother(text, 1100)
    blocks = SO.blockset_by_path_windows(g, 1100)
    p = S.default_params(local_alignment=0)
    for k in (0, 5):
        c = SO.collect(g, blocks[k])
        code, paths, cn = SO.poa(c, local=False)
        assert sm.block_graph_gfa(k, p, S.gpu_provider(engine)) == SO.to_gfa(SO.build_block_graph(c, code, paths, cn, ""))

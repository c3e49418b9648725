"""the reference-side plug-in point is the `tools` dictionary of RunBlast.run (uberBlast.py:327); every tool is
method(ref, qry) -> ndarray(object)[n, 15].  CPU test over the oracle-backed context: the product's public tools, held to that contract (tests/plugin_contract.py),
stacked and sent through the public object-row methods give the table RunBlast.run gives."""
import contextlib
import io
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _write_inputs(tmp_path, n=40):
    from peppan_amd import synth
    names, seqs = synth.make_genes(n, 0, seed=21)
    with open(tmp_path / 'g.fa', 'w') as f:
        for i, s in enumerate(seqs):
            f.write('>%d\n%s\n' % (i, s.decode()))
    return str(tmp_path / 'g.fa')


def test_public_tools_keep_the_reference_plugin_contract(tmp_path, monkeypatch):
    from oracle_context import OracleContext
    from plugin_contract import tools_then_methods
    from peppan_amd import uberBlast as UB
    fa = _write_inputs(tmp_path)
    octx = OracleContext()
    monkeypatch.setattr(UB, 'get_context', lambda device=None: octx)
    with contextlib.redirect_stderr(io.StringIO()):
        via_loop = tools_then_methods(UB.RunBlast(), fa, fa, ['blastn', 'diamond'], 0.4, 40., 0.25, re_score=1, fix_end=(3., 3.))
        via_run = UB.RunBlast().run(fa, fa, ['blastn', 'diamond'], 0.4, 40., 0.25, re_score=1, return_overlap=[False, 300, 0.6], fix_end=[3., 3.])
        argv_run = UB.uberBlast(('-r %s -q %s --blastn --diamond -s 1 --min_id 0.4 --min_cov 40 --min_ratio 0.25 -e 3,3' % (fa, fa)).split())
    assert via_loop.shape[0] > 80 and via_loop.shape[1] == 16
    assert via_loop.tolist() == via_run.tolist() == argv_run.tolist()
    # diamondSELF alone, with overlaps; and a tool without hits hands over an empty [0, 15] table instead of raising
    with contextlib.redirect_stderr(io.StringIO()):
        a, a_ovl = tools_then_methods(UB.RunBlast(), fa, fa, ['diamondSELF'], 0.4, 40., 0.25, fix_end=(0., 0.), overlap=(300, 0.6))
        b, b_ovl = UB.RunBlast().run(fa, fa, ['diamondSELF'], 0.4, 40., 0.25, return_overlap=[True, 300, 0.6], fix_end=[0., 0.])
    assert a.tolist() == b.tolist() and a_ovl.tolist() == b_ovl.tolist() and len(a) >= 40
    rb = UB.RunBlast()
    rb.min_id, rb.min_cov, rb.min_ratio, rb.table_id = 0.9999, 100000., 0.99, 11
    with contextlib.redirect_stderr(io.StringIO()):
        none = rb.runDiamond(fa, fa)
    assert isinstance(none, np.ndarray) and none.shape == (0, 15) and none.dtype == object


def test_subclass_tool_override_and_failed_tools(tmp_path, monkeypatch):
    """a subclass that replaces a tool (the Level-2 plug-in of INTEGRATION.md) is called by run(); a failing tool is reported, counted in
    failed_tools and the tables of the tools before it are kept (uberBlast.py:347-349)"""
    from oracle_context import OracleContext
    from peppan_amd import uberBlast as UB
    fa = _write_inputs(tmp_path, 24)
    octx = OracleContext()
    monkeypatch.setattr(UB, 'get_context', lambda device=None: octx)
    calls = []

    class WithPlugin(UB.RunBlast):
        def runDiamond(self, ref, qry, nhits=10, frames='7'):
            calls.append('plugin')
            return UB.RunBlast.runDiamond(self, ref, qry, nhits, frames)            # object rows, as the contract says

    with contextlib.redirect_stderr(io.StringIO()):
        plain = UB.RunBlast().run(fa, fa, ['diamond'], 0.4, 40., 0.25, return_overlap=[False, 300, 0.6])
        plug = WithPlugin().run(fa, fa, ['diamond'], 0.4, 40., 0.25, return_overlap=[False, 300, 0.6])
    assert calls == ['plugin'] and plug.tolist() == plain.tolist() and len(plain) > 20

    # diamondSELF goes through the instance's runDiamond (uberBlast.py:511): a subclass that overrides runDiamond only is not bypassed ...
    del calls[:]
    with contextlib.redirect_stderr(io.StringIO()):
        plain_self = UB.RunBlast().run(fa, fa, ['diamondSELF'], 0.4, 40., 0.25, return_overlap=[False, 300, 0.6])
        plug_self = WithPlugin().run(fa, fa, ['diamondSELF'], 0.4, 40., 0.25, return_overlap=[False, 300, 0.6])
    assert calls == ['plugin'] and plug_self.tolist() == plain_self.tolist() and len(plain_self) >= 24
    # ... and neither is a tool set on the INSTANCE (the reference's dictionary holds bound instance attributes, uberBlast.py:327)
    rb = UB.RunBlast()
    seen = []

    def my_blast(ref, qry):
        seen.append((ref, qry))
        return UB.RunBlast.runBlast(rb, ref, qry)
    rb.runBlast = my_blast
    with contextlib.redirect_stderr(io.StringIO()):
        inst = rb.run(fa, fa, ['blastn'], 0.4, 40., 0.25, return_overlap=[False, 300, 0.6])
        ref_b = UB.RunBlast().run(fa, fa, ['blastn'], 0.4, 40., 0.25, return_overlap=[False, 300, 0.6])
    assert seen == [(fa, fa)] and inst.tolist() == ref_b.tolist()

    class Broken(UB.RunBlast):
        def runDiamond(self, ref, qry, nhits=10, frames='7'):
            raise RuntimeError('tool fell over')

    rb = Broken()
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        only_blast = rb.run(fa, fa, ['blastn', 'diamond'], 0.4, 40., 0.25, return_overlap=[False, 300, 0.6])
        ref_blast = UB.RunBlast().run(fa, fa, ['blastn'], 0.4, 40., 0.25, return_overlap=[False, 300, 0.6])
    assert [m for m, _ in rb.failed_tools] == ['diamond'] and 'tool fell over' in err.getvalue()
    assert only_blast.tolist() == ref_blast.tolist() and len(ref_blast) >= 24
    assert UB.RunBlast().run(fa, fa, ['blastn'], 0.4, 40., 0.25) is not None and rb.failed_tools


def test_batch_entry_point_can_be_strict_about_failed_tools(tmp_path, monkeypatch):
    """uberBlastBatch(strict=True) - what the genome mapping calls - raises when a tool fails instead of returning the other tools' hits: stores
    written from half a search would pass for results.  Without it the reference's convention holds (reported, left out)."""
    import pytest
    from oracle_context import OracleContext
    from peppan_amd import uberBlast as UB
    fa = _write_inputs(tmp_path, 12)
    octx = OracleContext()
    monkeypatch.setattr(UB, 'get_context', lambda device=None: octx)

    def broken(self, ref, qry):
        raise TypeError('a tool that falls over')
    monkeypatch.setattr(UB.RunBlast, '_runBlast_table', broken)
    monkeypatch.setitem(UB._BUILTIN_TOOLS, 'runBlast', None)                     # (run() then goes through the method)
    argv = ('-q %s --blastn -s 1 --min_id 0.6 --min_cov 50 --min_ratio 0.2 -e 0,3' % fa).split()
    refs = [[('c1', 'ACGT' * 200)]]
    with contextlib.redirect_stderr(io.StringIO()):
        lenient = UB.uberBlastBatch(refs, argv)
        assert len(lenient) == 1 and len(lenient[0]) == 0
        with pytest.raises(RuntimeError, match='blastn failed'):
            UB.uberBlastBatch(refs, argv, strict=True)

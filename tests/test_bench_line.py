"""bench.py's stdout contract: ONE compact line the driver can parse (round 5's 25 KB line was not: BENCH_r05.json "parsed": null).
The line builder is a pure function of the detail record; it is run here on canned statistics - round 5's full record, which is tracked
under profiles/ - and on a record padded with long strings and non-finite numbers."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _fail(name):
    raise AssertionError('non-finite constant %s in the bench line' % name)


def _canned():
    text = open(os.path.join(ROOT, 'profiles', 'r05_bench_line.txt')).read().strip().splitlines()[-1]
    return json.loads(text)


def _check(line):
    assert '\n' not in line
    assert len(line.encode()) < 4096
    d = json.loads(line, parse_constant=_fail)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(d['roofline'])
    assert d['roofline']['bound'] in ('hbm', 'mfma')
    assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(d['cpu_baseline'])
    assert 'workload' in d['config'] and 'model' not in d['config']

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                for s in strings(v):
                    yield s
        elif isinstance(o, list):
            for v in o:
                for s in strings(v):
                    yield s
        elif isinstance(o, str):
            yield o
    assert all(len(s) <= 120 for s in strings(d))
    return d


def test_compact_line_of_round5_statistics():
    detail = _canned()
    assert len(json.dumps(detail)) > 20000                    # (the record that was too long as a line)
    d = _check(bench.compact_line(detail))
    assert d['metric'] == 'gene_pairs_aligned_per_s' and d['n_gpus'] == 1
    assert abs(d['value'] - detail['value']) < 1e-5 * detail['value']
    assert abs(d['roofline']['frac'] - detail['roofline']['frac']) < 1e-5
    assert d['roofline']['kernel'] == detail['roofline']['kernel']
    assert d['cpu_baseline']['gpu_hits_identical'] is True
    # one scalar per secondary leg
    for k in ('north_star_call_ms', 'uberblast_e2e_ms', 'get_similar_pairs_ms', 'search_50k_ms', 'search_10k_blastn_ms', 'map_genomes_per_s', 'pool_genomes_per_s'):
        assert isinstance(d[k], float), k


def test_compact_line_survives_padding_and_non_finite_numbers():
    detail = _canned()
    detail['config']['workload'] = 'w' * 5000
    detail['config']['parallelism'] = 'p' * 5000
    detail['cpu_baseline']['sample'] = 's' * 5000
    detail['cpu_baseline']['cpu_model'] = 'c' * 5000
    detail['roofline']['kernel'] = 'k' * 5000
    detail['roofline']['traffic'] = float('nan')
    detail['sw_cell_updates_per_s_per_gpu'] = float('inf')
    detail['workloads'] = {'error': 'x' * 100000}
    d = _check(bench.compact_line(detail))
    assert d['roofline']['traffic'] is None and d['sw_cell_updates_per_s_per_gpu'] is None
    assert d['leg_errors'] == ['workloads']


def test_compact_line_without_optional_parts():
    detail = {k: v for k, v in _canned().items() if k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                                           'vs_baseline', 'dtype', 'data', 'config')}
    line = bench.compact_line(detail)
    d = json.loads(line, parse_constant=_fail)
    assert d['roofline'] is None and d['cpu_baseline'] is None and d['parity_check'] is None
    assert len(line) < 1500


def test_emit_writes_detail_and_prints_one_line(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    monkeypatch.setattr(bench, 'DETAIL_FILE', None)
    bench.emit(_canned())
    out = capsys.readouterr().out
    assert out.endswith('\n') and out.count('\n') == 1
    _check(out.strip())
    full = json.load(open(os.path.join(str(tmp_path), 'bench_detail.json')))
    assert 'roofline_kernels' in full and 'workloads' in full


def test_compact_line_with_a_valu_bound_kernel_on_top():
    """small workloads: a Smith-Waterman pass is the largest kernel - the line's roofline stays the contract's HBM one (its HBM side), the issue fraction beside it"""
    detail = _canned()
    detail['roofline'] = [e for e in detail['roofline_kernels'] if e['bound'] == 'valu'][0]
    d = _check(bench.compact_line(detail))
    assert d['roofline']['unit'] == 'GB/s' and d['roofline']['peak'] == 8000.0 and 0 < d['roofline']['frac'] < 0.05
    assert 0.5 < d['roofline_valu_issue_frac'] <= 1.01

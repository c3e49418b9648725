"""What the clusterer (K9 = oracle_linclust, bit for bit) delivers against an exhaustive pairwise pass - a floor that is checked every round
(tests/recall_linclust.py holds the method; full runs are tracked in profiles/r05_recall_linclust.jsonl)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_iterative_clustering_brings_every_close_pair_under_one_exemplar():
    """alleles of 50 genes - identical copies, substitution alleles, indel alleles, shuffled - through the 11 levels of iterClust's schedule
    (PEPPAN.py:1777-1792; identity 0.9, coverage 0.8 = PEPPAN.py:1692-1693): all pairs aligned exhaustively (oracle/full_sw.c).  A single level may
    leave close pairs apart (only the centre of a k-mer group is compared, as in linclust); the schedule as a whole must not."""
    import recall_linclust as R
    r = R.report(R.synth_instances(50, 4, seed=21))
    assert r['input_pairs_within_last_level'] >= 150 and len(r['levels']) == 11
    assert r['final_recall'] >= 0.99, r
    assert r['truth_pairs_left_after_last_level'] == 0, r
    assert r['recall_all_levels'] >= 0.8, r                                  # (per level: what a level sees and settles itself)
    assert 30 <= r['final_exemplars'] <= 50, r                               # (the 50 genes come in families of four, the two closest members 5 % apart: ~37 exemplars) nothing merged that should not be
    assert all(l['recall'] is None or l['recall'] >= 0.7 for l in r['levels']), r['levels']

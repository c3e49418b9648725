"""Test helper: the reference's OWN driver loop over the tools of a RunBlast object, restated line by line from
/root/reference/modules/uberBlast.py:327, 343-376 - what a PEPPAN maintainer's unmodified `run` does with whatever a tool
registered in the `tools` dictionary returns (method(ref, qry) -> ndarray(object)[n, 15]).  The product's
public runBlast / runDiamond / runDiamondSELF must survive exactly this: `b.shape[0]`, np.vstack, np.hstack with an arange
column, then the object-row post-processing methods and the pandas sort."""
import numpy as np
import pandas as pd


def reference_style_run(rb, ref, qry, methods, min_id, min_cov, min_ratio, table_id=11, re_score=0, fix_end=(6., 6.), return_overlap=(False, 300, 0.6)):
    tools = dict(blastn=rb.runBlast, diamond=rb.runDiamond, diamondself=rb.runDiamondSELF)          # uberBlast.py:327
    rb.min_id, rb.min_cov, rb.min_ratio, rb.table_id = min_id, min_cov, min_ratio, table_id         # :328-332
    blastab = []
    for method in methods:                                                                           # :343-345
        if method.lower() in tools:
            blastab.append(tools[method.lower()](ref, qry))
    for b in blastab:
        assert isinstance(b, np.ndarray) and b.dtype == object and b.ndim == 2 and b.shape[1] == 15, (type(b), getattr(b, 'shape', None))
    blastab = [b for b in blastab if b.shape[0] > 0]                                                 # :346
    if not blastab:
        return np.empty([0, 16], dtype=object)
    blastab = np.vstack(blastab)                                                                     # :353
    blastab = np.hstack([blastab, np.arange(blastab.shape[0], dtype=int)[:, np.newaxis]])            # :354
    for row in blastab[:50]:
        # the types parseDiamond / parseBlast produce (uberBlast.py:57-58, 280-288): names str, CIGAR [[n, op], ...]
        assert isinstance(row[0], str) and isinstance(row[1], str) and isinstance(row[14], list) and isinstance(row[14][0][0], int) and row[14][0][1] in 'MID'
    if re_score:
        blastab = rb.reScore(ref, qry, blastab, re_score, rb.min_id, rb.table_id)                    # :363-364
    rb.fixEnd(blastab, *fix_end)                                                                     # :369
    if return_overlap[0]:
        overlap = rb.returnOverlap(blastab, list(return_overlap))                                    # :370-373
        return pd.DataFrame(blastab).sort_values([0, 1, 11]).values, overlap
    return pd.DataFrame(blastab).sort_values([0, 1, 11]).values                                      # :375

"""synthgenes-v1: seeded synthetic gene sets of the sizes BASELINE.json names.

Ancestral CDS = ATG + uniform sense codons + TAA.  Genes come in super-families of 4 derived from one root
by i.i.d. nucleotide substitution at rates {0, 0.05, 0.15, 0.35} (codons that become stops are restored)
plus codon indels at 1/300 per codon with geometric length (mean 2).  Names are integers, as PEPPAN's
encodeNames produces (PEPPAN.py:1766-1775)."""
import numpy as np

_B = np.frombuffer(b'ACGT', dtype=np.uint8)
_SENSE = np.array([c for c in range(64) if c not in (48, 50, 56)], dtype=np.int64)   # A0 C1 G2 T3: TAA=48 TAG=50 TGA=56
SUB_RATES = (0.0, 0.05, 0.15, 0.35)


def _codons_to_bytes(cod):
    out = np.empty(cod.size * 3, dtype=np.uint8)
    out[0::3] = _B[cod >> 4]
    out[1::3] = _B[(cod >> 2) & 3]
    out[2::3] = _B[cod & 3]
    return out


def _derive(rng, root, rate):
    cod = root.copy()
    n = cod.size
    if rate > 0:
        inner = slice(1, n - 1)
        b = np.stack([cod >> 4, (cod >> 2) & 3, cod & 3], 1)
        mut = rng.random(b.shape) < rate
        mut[0] = mut[-1] = False
        nb = np.where(mut, rng.integers(0, 4, b.shape), b)
        nc = (nb[:, 0] << 4) | (nb[:, 1] << 2) | nb[:, 2]
        stop = (nc == 48) | (nc == 50) | (nc == 56)
        stop[0] = stop[-1] = False
        cod = np.where(stop, cod, nc)
        del inner
    # codon indels, 1/300 per codon, geometric length mean 2, never touching the first / last 2 codons
    ev = np.nonzero(rng.random(n) < (1.0 / 300.0))[0]
    ev = ev[(ev > 2) & (ev < n - 4)]
    for p in ev[::-1]:
        k = int(rng.geometric(0.5))
        if rng.random() < 0.5:
            cod = np.concatenate([cod[:p], cod[min(p + k, cod.size - 2):]])
        else:
            cod = np.concatenate([cod[:p], _SENSE[rng.integers(0, _SENSE.size, k)], cod[p:]])
    return cod


def gene_lengths(rng, n, fixed=1002):
    if fixed:
        return np.full(n, fixed // 3, dtype=np.int64)
    L = np.exp(rng.normal(6.65, 0.55, n))
    L = np.clip((np.round(L / 3) * 3).astype(np.int64), 120, 9492)
    return L // 3


def make_genes(n_genes, fixed_len=1002, seed=355, family=4):
    """returns (names list[str], seqs list[bytes]) with n_genes genes in families of `family`"""
    rng = np.random.default_rng(seed)
    n_fam = (n_genes + family - 1) // family
    ncod = gene_lengths(rng, n_fam, fixed_len)
    seqs = []
    for f in range(n_fam):
        root = np.concatenate([[14], _SENSE[rng.integers(0, _SENSE.size, int(ncod[f]) - 2)], [48]])   # ATG ... TAA
        for k in range(family):
            if len(seqs) == n_genes:
                break
            seqs.append(_codons_to_bytes(_derive(rng, root, SUB_RATES[k % len(SUB_RATES)])).tobytes())
    names = [str(i) for i in range(len(seqs))]
    return names, seqs


def make_proteins(n, length=300, seed=1, family=3, sub=0.2, indel=0.7):
    """random protein families as residue-code arrays (letter - 'A'), for kernel-level tests; indel = probability that a member carries one
    insertion or deletion of 1-8 residues"""
    rng = np.random.default_rng(seed)
    aa = np.frombuffer(b'ARNDCQEGHILKMFPSTWYV', dtype=np.uint8) - 65
    out = []
    while len(out) < n:
        L = int(length if np.isscalar(length) else rng.integers(length[0], length[1]))
        root = aa[rng.integers(0, 20, L)]
        out.append(root.copy())
        for _ in range(family - 1):
            m = root.copy()
            pos = rng.random(L) < sub
            m[pos] = aa[rng.integers(0, 20, int(pos.sum()))]
            if L > 60 and rng.random() < indel:
                p = int(rng.integers(20, L - 30)); k = int(rng.integers(1, 9))
                m = np.concatenate([m[:p], m[p + k:]]) if rng.random() < 0.5 else np.concatenate([m[:p], aa[rng.integers(0, 20, k)], m[p:]])
            out.append(m)
    return [np.ascontiguousarray(x, dtype=np.uint8) for x in out[:n]]


PAN_GENOME_PRESENCE = ((0.99, 0.5, 0.05), (0.40, 0.22, 0.38))    # (presence probabilities, shares of the families): core / shell / cloud of a
#                                                                    50 000-gene pan-genome (12 500 families) whose genomes carry about 6 500 genes, 7 Mb, each (BASELINE configs[4], "Salmonella-scale")


def make_genomes(gene_seqs, n_genomes, seed=355, family=4, presence=None):
    """synthgenes-v1 genomes: a family is present with p = 0.99 (70 % of families), 0.5 (20 %) or 0.05
    (10 %) - or with the probabilities / shares given as presence = ((p...), (share...)), e.g. PAN_GENOME_PRESENCE; the allele is one
    family member with a per-genome substitution rate U(0, 0.02); one contig, random strand, 50-300 nt random spacers.
    Returns [(genome name, contig bytes, [(gene index, start, end, strand), ...]), ...]"""
    rng = np.random.default_rng(seed + 1)
    n_fam = (len(gene_seqs) + family - 1) // family
    probs, shares = presence or ((0.99, 0.5, 0.05), (0.7, 0.2, 0.1))
    p_present = rng.choice(list(probs), size=n_fam, p=list(shares))
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b'ACGT')] = list(b'TGCA')
    out = []
    for g in range(n_genomes):
        rate = rng.uniform(0, 0.02)
        parts, ann, pos = [], [], 0
        for f in np.nonzero(rng.random(n_fam) < p_present)[0]:
            k = int(f) * family + int(rng.integers(0, min(family, len(gene_seqs) - int(f) * family)))
            s = np.frombuffer(gene_seqs[k], dtype=np.uint8).copy()
            m = rng.random(s.size) < rate
            m[:3] = m[-3:] = False
            s[m] = _B[rng.integers(0, 4, int(m.sum()))]
            sp = _B[rng.integers(0, 4, int(rng.integers(50, 301)))]
            strand = '+' if rng.random() < 0.5 else '-'
            if strand == '-':
                s = comp[s[::-1]]
            parts += [sp, s]
            pos += sp.size
            ann.append((k, pos + 1, pos + s.size, strand))
            pos += s.size
        parts.append(_B[rng.integers(0, 4, 100)])
        out.append(('g%04d' % g, np.concatenate(parts).tobytes(), ann))
    return out


def make_instances(n_base, copies, seed=8):
    """gene instances of a pan-genome for the front end (writeGenes / iterClust): n_base genes (log-normal lengths), `copies` alleles each -
    60 % identical to the gene, the rest one of four variants with one to three substitutions.  Returns a list of n_base * copies strings."""
    rng = np.random.default_rng(seed)
    names, seqs = make_genes(n_base, 0, seed=seed)
    out = []
    for s in seqs:
        a = np.frombuffer(s, dtype=np.uint8)
        variants = [s.decode()]
        for _ in range(4):
            v = a.copy()
            pos = rng.integers(3, len(v) - 3, int(rng.integers(1, 4)))
            v[pos] = _B[rng.integers(0, 4, len(pos))]
            variants.append(v.tobytes().decode())
        out += [variants[k] for k in rng.choice(5, size=copies, p=[0.6, 0.1, 0.1, 0.1, 0.1]).tolist()]
    return out

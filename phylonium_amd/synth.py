"""Deterministic synthetic genomes for tests and benchmarks.

The reference's own generator (test/simf.cxx) only makes substitution-only star
phylogenies whose bytes depend on libstdc++ internals.  This one is
numpy-only (PCG64, fixed algorithms), and adds what BASELINE.json's configs
3-5 need: indels, inverted blocks (so revseqcmp runs), multiple contigs (so
'!' appears) and tree-like descent.  Host-side tooling, not on the hot path.
"""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[list(b"ACGT")] = list(b"TGCA")


def jc_to_p(d):
    """JC distance → expected fraction of differing sites (test/simf.cxx:62-68)."""
    return 0.75 - 0.75 * np.exp(-(4.0 / 3.0) * d)


def random_base(length, rng):
    return ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]


def mutate(seq, p, rng):
    """Substitute each site independently with probability p by a different base."""
    out = seq.copy()
    hit = np.flatnonzero(rng.random(seq.size) < p)
    if hit.size:
        code = np.searchsorted(ACGT, out[hit]).astype(np.uint8)  # ACGT is sorted
        out[hit] = ACGT[(code + rng.integers(1, 4, size=hit.size, dtype=np.uint8)) & 3]
    return out


def revcomp(seq):
    return _COMP[seq[::-1]]


def structural(seq, rng, indel_events=0, indel_max=50, inv_events=0, inv_len=(200, 2000)):
    """Apply indel and inversion events; returns a new array."""
    pieces = []
    n = seq.size
    ev = []
    for _ in range(indel_events):
        ev.append((int(rng.integers(0, n)), "indel"))
    for _ in range(inv_events):
        ev.append((int(rng.integers(0, n)), "inv"))
    ev.sort()
    pos = 0
    for at, kind in ev:
        if at < pos:
            continue
        pieces.append(seq[pos:at])
        if kind == "indel":
            ln = int(rng.integers(1, indel_max + 1))
            if rng.random() < 0.5:  # deletion
                pos = min(n, at + ln)
            else:  # insertion of random bases
                pieces.append(random_base(ln, rng))
                pos = at
        else:
            ln = int(rng.integers(inv_len[0], inv_len[1] + 1))
            end = min(n, at + ln)
            pieces.append(revcomp(seq[at:end]))
            pos = end
    pieces.append(seq[pos:])
    return np.concatenate(pieces) if pieces else seq.copy()


def split_contigs(seq, n_contigs, rng):
    """Cut into contigs joined by '!' (src/sequence.cxx:171-199)."""
    if n_contigs <= 1:
        return seq
    cuts = np.sort(rng.choice(np.arange(1, seq.size), size=n_contigs - 1, replace=False))
    parts = np.split(seq, cuts)
    out = []
    for k, p in enumerate(parts):
        if k:
            out.append(np.frombuffer(b"!", dtype=np.uint8))
        out.append(p)
    return np.concatenate(out)


def make_genomes(n, length, seed=1, d_range=(0.01, 0.3), tree=False, indel_per_mbp=0.0,
                 inv_frac=0.0, contigs=1, inv_len=(200, 2000)):
    """n genomes of ~`length` bases as uint8 arrays over {A,C,G,T,!}.

    star (tree=False): genome g = base mutated at JC distance d_g ~ U(d_range).
    tree=True: genome g descends from genome (g-1)//2 (heap order) at a branch
    distance d_g/4, so pair distances span a range.
    """
    rng = np.random.default_rng(seed)
    base = random_base(length, rng)
    out = []
    for g in range(n):
        d = float(rng.uniform(*d_range))
        if tree:
            if g == 0:
                s = base.copy()
            else:
                parent = out[(g - 1) // 2]
                parent = parent[parent != ord("!")]
                s = mutate(parent, jc_to_p(d / 4), rng)
        else:
            s = mutate(base, jc_to_p(d), rng)
        n_indel = int(round(indel_per_mbp * s.size / 1e6))
        mean_inv = 0.5 * (inv_len[0] + inv_len[1])
        n_inv = int(round(inv_frac * s.size / mean_inv))
        if n_indel or n_inv:
            s = structural(s, rng, indel_events=n_indel, inv_events=n_inv, inv_len=inv_len)
        s = split_contigs(s, contigs, rng)
        out.append(np.ascontiguousarray(s))
    return out

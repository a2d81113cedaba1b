"""Seeded synthetic genomes for tests and bench.py.

No real E. coli / RefSeq data exists in the container, so every configuration
is generated: uniform i.i.d. ACGT ancestors, descendants by independent point
substitutions at rate mu, FASTA lines of width 70.  Unrelated random genomes
share ~no k-mers, so the family structure is what makes the comparator work.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_genome(rng, length):
    """uint8 array of ASCII A/C/G/T."""
    return _ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]


def mutate(rng, genome, mu):
    """independent substitutions at rate mu (always to a different base)."""
    g = genome.copy()
    if mu <= 0:
        return g
    n_mut = rng.binomial(len(g), mu)
    if n_mut == 0:
        return g
    pos = rng.choice(len(g), size=n_mut, replace=False)
    idx = np.searchsorted(_ACGT, g[pos])  # A,C,G,T are already sorted in ASCII
    g[pos] = _ACGT[(idx + rng.integers(1, 4, size=n_mut)) % 4]
    return g


def family_genomes(seed, n_genomes, length, n_families, mus, length_jitter=0.0):
    """n_genomes genomes in n_families families; member j of a family is the
    ancestor mutated at mus[j % len(mus)] (member 0 of each family at mus[0])."""
    rng = np.random.default_rng(seed)
    per = (n_genomes + n_families - 1) // n_families
    out = []
    for f in range(n_families):
        L = length if not length_jitter else int(length * (1.0 + length_jitter * (rng.random() * 2 - 1)))
        anc = random_genome(rng, L)
        for j in range(per):
            if len(out) >= n_genomes:
                break
            out.append(mutate(rng, anc, mus[j % len(mus)]))
    return out


def to_fasta(genome, name="g", n_records=1, width=70):
    """uint8 genome -> FASTA bytes with n_records records, line width 70."""
    n = len(genome)
    cuts = [n * i // n_records for i in range(n_records + 1)]
    parts = []
    for r in range(n_records):
        seg = genome[cuts[r]:cuts[r + 1]]
        parts.append((">%s_%d\n" % (name, r)).encode())
        full = (len(seg) // width) * width
        if full:
            body = np.empty((len(seg) // width, width + 1), dtype=np.uint8)
            body[:, :width] = seg[:full].reshape(-1, width)
            body[:, width] = 10
            parts.append(body.tobytes())
        if len(seg) > full:
            parts.append(seg[full:].tobytes() + b"\n")
    return b"".join(parts)


def concat_records(genomes):
    """list of uint8 arrays -> (bases, rec_off) as the C-ABI wants them."""
    off = np.zeros(len(genomes) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(g) for g in genomes])
    return (np.concatenate(genomes) if genomes else np.zeros(0, np.uint8)), off

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


class DirectSketches:
    """Sketches synthesised directly at the super-k-mer level (SURVEY.md 8d allows it for the comparator configs): the
    sorted distinct (minimizer, canonical k-mer) keys of every sketch, ready for spsp_compare_device, plus the stored
    super-k-mers they were rolled from, so that any sketch can also be written out in the on-disk sketch format
    (`payload`) for a reader that starts from files (the oracle, spsp_sketch_decode_device)."""

    def __init__(self, k, m, minimizer, kmer_lo, sk_off, universe, skm_off, skm_bucket, skm_flank):
        self.k, self.m = k, m
        self.minimizer, self.kmer_lo, self.sk_off = minimizer, kmer_lo, sk_off     # int32 / int64 tensors (bit patterns), np.uint64[n + 1]
        self.universe = universe                                                   # int64 tensor: the selectable minimizer values, ascending
        self.skm_off, self.skm_bucket, self.skm_flank = skm_off, skm_bucket, skm_flank

    @property
    def n(self):
        return len(self.sk_off) - 1

    def payload(self, i, rate=1000.0):
        """sketch i as sub_sampler would have written it (gunzipped): header, then per bucket in ascending minimizer
        order [m ASCII][u32 n][blob of the maximal super-k-mers' k-m prefix and suffix bases, 4 per byte]["\\n\\n"]"""
        k, m, side = self.k, self.m, self.k - self.m
        r0, r1 = int(self.skm_off[i]), int(self.skm_off[i + 1])
        bucket = self.skm_bucket[r0:r1].cpu().numpy()
        flank = self.skm_flank[r0:r1].cpu().numpy()
        uni = self.universe.cpu().numpy()
        out = [b"%d %d %d %f\n" % (2 * k - m, m, (r1 - r0) * (side + 1), rate)]
        order = np.argsort(bucket, kind="stable")
        bucket, flank = bucket[order], flank[order]
        nuc = np.frombuffer(b"ACTG", dtype=np.uint8)                               # int2nuc: A=0 C=1 T=2 G=3
        cuts = np.flatnonzero(np.diff(bucket)) + 1
        for lo, hi in zip(np.concatenate([[0], cuts]), np.concatenate([cuts, [len(bucket)]])):
            if hi == lo:
                continue
            mn = int(uni[bucket[lo]])
            out.append(nuc[[(mn >> (2 * (m - 1 - j))) & 3 for j in range(m)]].tobytes())
            codes = flank[lo:hi].reshape(-1, 4).astype(np.uint8)                   # 2 (k - m) is a multiple of 4 for k, m odd
            blob = bytes([0]) + ((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).astype(np.uint8).tobytes()
            out.append(len(blob).to_bytes(4, "little") + blob + b"\n\n")
        return b"".join(out)


def direct_family_sketches(n, fam_size=20, k=31, m=11, seed=4, device="cpu", skm_range=(120, 480), mus=(0.001, 0.01, 0.05),
                           n_buckets=100):
    """n sketches in families of fam_size (BASELINE configs[2] / configs[3] shape): a family's ancestor is a set of
    maximal super-k-mers (k-m random bases either side of a minimizer drawn from a universe of n_buckets values -- at
    k=31 m=11 s=1000 only ~100 minimizer values are selectable at all); member j carries every ancestral
    super-k-mer with independent substitutions at rate mus[j % len(mus)] in the flanks, and loses it (a fresh random one
    takes its place) when a substitution falls into the minimizer.  Each super-k-mer contributes its k-m+1 canonical
    k-mers; a sketch = the sorted distinct (minimizer, k-mer) keys.  torch on `device` (plumbing: setup is untimed)."""
    import torch
    assert k <= 32 and (k - m) % 2 == 0 and m <= 15
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    side = k - m
    n_fam = (n + fam_size - 1) // fam_size
    universe = torch.unique(torch.randint(0, 4 ** m, (4 * n_buckets,), generator=g, device=dev, dtype=torch.int64))[:n_buckets]
    n_buckets = int(universe.numel())
    h_fam = torch.randint(skm_range[0], skm_range[1] + 1, (n_fam,), generator=g, device=dev, dtype=torch.int64)
    fam_off = torch.cumsum(h_fam, 0) - h_fam
    total_anc = int(h_fam.sum().item())
    anc_bucket = torch.randint(0, n_buckets, (total_anc,), generator=g, device=dev, dtype=torch.int64)
    anc_flank = torch.randint(0, 4, (total_anc, 2 * side), generator=g, device=dev, dtype=torch.uint8)
    sk = torch.arange(n, device=dev)
    h_sk = h_fam[sk // fam_size]
    skm_off = torch.cumsum(h_sk, 0) - h_sk
    total = int(h_sk.sum().item())
    sk_of = torch.repeat_interleave(sk, h_sk)
    row = fam_off[sk_of // fam_size] + (torch.arange(total, device=dev) - skm_off[sk_of])
    mu = torch.tensor(mus, device=dev, dtype=torch.float32)[(sk_of % fam_size) % len(mus)]
    bucket, flank = anc_bucket[row], anc_flank[row]
    hit = torch.rand((total, 2 * side), generator=g, device=dev) < mu[:, None]
    flank = torch.where(hit, (flank + torch.randint(1, 4, flank.shape, generator=g, device=dev, dtype=torch.uint8)) % 4, flank)
    lost = torch.rand((total,), generator=g, device=dev) < 1.0 - (1.0 - mu) ** m
    flank = torch.where(lost[:, None], torch.randint(0, 4, flank.shape, generator=g, device=dev, dtype=torch.uint8), flank)
    bucket = torch.where(lost, torch.randint(0, n_buckets, (total,), generator=g, device=dev, dtype=torch.int64), bucket)
    del hit, lost, row, mu
    mn_val = universe[bucket]
    shifts = 2 * (m - 1 - torch.arange(m, device=dev))
    codes = torch.cat([flank[:, :side], ((mn_val[:, None] >> shifts[None, :]) & 3).to(torch.uint8), flank[:, side:]], dim=1)   # [total, 2k - m]
    w = side + 1
    fwd = torch.zeros((total, w), dtype=torch.int64, device=dev)
    rev = torch.zeros((total, w), dtype=torch.int64, device=dev)
    for t in range(k):                                    # k-mer starting at p: base t is codes[p + t]
        c = codes[:, t:t + w].to(torch.int64)
        fwd = (fwd << 2) | c
        rev = rev | ((c ^ 2) << (2 * t))                  # complement, reversed: base t lands 2t bits up
    lo = torch.minimum(fwd, rev).reshape(-1)              # (both < 2^62: signed order = unsigned order)
    del fwd, rev, codes
    key_hi = ((sk_of << 32) | mn_val).repeat_interleave(w)
    lo, order = torch.sort(lo, stable=True)
    key_hi = key_hi[order]
    key_hi, order = torch.sort(key_hi, stable=True)
    lo = lo[order]
    del order
    first = torch.ones(lo.numel(), dtype=torch.bool, device=dev)
    first[1:] = (lo[1:] != lo[:-1]) | (key_hi[1:] != key_hi[:-1])
    lo, key_hi = lo[first], key_hi[first]
    cnt = torch.bincount(key_hi >> 32, minlength=n).cpu().numpy()
    sk_off = np.zeros(n + 1, dtype=np.uint64)
    sk_off[1:] = np.cumsum(cnt)
    skm_off_np = np.zeros(n + 1, dtype=np.int64)
    skm_off_np[1:] = np.cumsum(h_sk.cpu().numpy())
    return DirectSketches(k, m, (key_hi & 0xffffffff).to(torch.int32).contiguous(), lo.contiguous(), sk_off, universe, skm_off_np,
                          bucket, flank)


def device_family_batches(n_genomes, seed, device, fam_size=20, len_range=(2_000_000, 8_000_000), mus=(0.001, 0.01, 0.05),
                          genomes_per_batch=100):
    """BASELINE configs[2]'s true shape (SURVEY.md 8d "C3": N genomes, L ~ U[2, 8] Mbp, families of 20, mu in
    {0.001, 0.01, 0.05}), generated ON THE DEVICE batch by batch so that 5 Gbp never exist on the host: a family's ancestor
    is uniform i.i.d. ACGT of a length drawn per family, member j is the ancestor under independent substitutions at
    mus[j % len(mus)] (always to a different base); genome i has 1 + i % 3 records.  Yields per batch a dict with
    `bases` (uint8 torch tensor on `device`: cleaned ASCII records back to back + 64 bytes of padding), `rec_off` (np.uint64,
    records + 1), `first_rec` (np.uint32, genomes + 1: genome g = records [first_rec[g], first_rec[g + 1])), `lengths`
    and `first_genome`.  torch is plumbing here (device memory + the random generator), as in bench.py."""
    import torch
    assert genomes_per_batch % fam_size == 0 and n_genomes % fam_size == 0
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    host_rng = np.random.default_rng(seed)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    for g0 in range(0, n_genomes, genomes_per_batch):
        nb = min(genomes_per_batch, n_genomes - g0)
        fams = nb // fam_size
        lens = [int(host_rng.integers(len_range[0], len_range[1] + 1)) for _ in range(fams)]
        total = sum(L * fam_size for L in lens)
        bases = torch.empty(total + 64, dtype=torch.uint8, device=device)
        bases[total:] = 65
        rec_off, first_rec, lengths = [0], [0], []
        at = 0
        for f, L in enumerate(lens):
            anc = torch.randint(0, 4, (L,), device=device, generator=gen, dtype=torch.int64)
            for j in range(fam_size):
                mu = mus[j % len(mus)]
                hit = torch.rand(L, device=device, generator=gen) < mu
                shift = torch.randint(1, 4, (L,), device=device, generator=gen, dtype=torch.int64)
                bases[at:at + L] = lut[(anc + hit * shift) % 4]
                i = g0 + f * fam_size + j
                nr = 1 + i % 3
                for r in range(nr):
                    rec_off.append(at + L * (r + 1) // nr)
                first_rec.append(len(rec_off) - 1)
                lengths.append(L)
                at += L
            del anc
        yield {"bases": bases, "n_bases": total, "rec_off": np.asarray(rec_off, dtype=np.uint64), "first_rec": np.asarray(first_rec, dtype=np.uint32),
               "lengths": lengths, "first_genome": g0}

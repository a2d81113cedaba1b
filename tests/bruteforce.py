"""First-principle definitions used to pin the oracle (NOT derived from the
reference's code and not from oracle/): python-xxhash for XXH64, plain Python
for 2-bit codes, reverse complements, window minima and set intersections.

SURVEY.md 8(c) "Independent checks that replace a runnable oracle" (1)-(5).
"""
import struct

import numpy as np
import xxhash

CODE = {"A": 0, "C": 1, "T": 2, "G": 3}   # (c/2)%4, reference utils.cpp:13-16
NUC = "ACTG"
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def val(s):
    v = 0
    for c in s:
        v = (v << 2) | CODE[c]
    return v


def to_str(v, n):
    return "".join(NUC[(v >> (2 * (n - 1 - i))) & 3] for i in range(n))


def rc_str(s):
    return "".join(COMP[c] for c in reversed(s))


def canon_val(s):
    return min(val(s), val(rc_str(s)))


def h64(x, seed=1312):
    return xxhash.xxh64_intdigest(struct.pack("<Q", x), seed=seed)


def random_dna(rng, n):
    return "".join(np.array(list("ACGT"))[rng.integers(0, 4, size=n)])


def mutate(rng, s, mu):
    a = np.array(list(s))
    hit = np.nonzero(rng.random(len(a)) < mu)[0]
    for p in hit:
        a[p] = "ACGT"[("ACGT".index(a[p]) + int(rng.integers(1, 4))) % 4]
    return "".join(a)


def fasta(records, width=70, names=None):
    out = []
    for i, r in enumerate(records):
        out.append(">" + (names[i] if names else "rec%d" % i))
        for p in range(0, len(r), width):
            out.append(r[p:p + width])
    return ("\n".join(out) + "\n").encode()


def mmer_hashes(seq, m):
    """canonical value and hash of every m-mer of seq."""
    cv = [canon_val(seq[p:p + m]) for p in range(len(seq) - m + 1)]
    return cv, [h64(x) for x in cv]


def selected_kmers(seq, k, m, T):
    """list of (j, bucket) for every k-mer whose window minimum hash <= T."""
    if len(seq) < k:
        return []
    cv, hs = mmer_hashes(seq, m)
    w = k - m + 1
    out = []
    for j in range(len(seq) - k + 1):
        best = min(range(j, j + w), key=lambda p: hs[p])
        if hs[best] <= T:
            out.append((j, cv[best]))
    return out


def sketch_set(records, k, m, T):
    """{(bucket minimizer value, canonical k-mer value)} over all records."""
    s = set()
    for seq in records:
        for j, b in selected_kmers(seq, k, m, T):
            s.add((b, canon_val(seq[j:j + k])))
    return s


def parse_payload(payload):
    """Uncompressed sketch payload -> (header fields, {minimizer_str: [super-k-mer strings]}, order).

    Format per SURVEY.md 8(b): header line "<2k-m> <m> <n> <rate>\\n", then per
    bucket [m ASCII][u32 LE nbytes][blob][(prefix\\nsuffix\\n)*]["\\n\\n"].
    Blob: byte0 = len%4, then 4 bases/byte MSB first (A=0 C=1 T=2 G=3).
    """
    nl = payload.index(b"\n")
    f = payload[:nl].split(b" ")
    skm, m = int(f[0]), int(f[1])
    k = (skm + m) // 2
    hdr = {"skmer": skm, "m": m, "k": k, "n": int(f[2]), "rate": f[3].decode()}
    pos = nl + 1
    buckets = {}
    order = []
    while pos < len(payload):
        mini = payload[pos:pos + m].decode(); pos += m
        nb = struct.unpack_from("<I", payload, pos)[0]; pos += 4
        blob = payload[pos:pos + nb]; pos += nb
        seqs = []
        if nb:
            mod = blob[0]
            assert mod == 0, "sketch blobs always hold a multiple of 4 bases"
            bases = "".join(NUC[(b >> sh) & 3] for b in blob[1:] for sh in (6, 4, 2, 0))
            half = k - m
            assert len(bases) % (2 * half) == 0
            for i in range(0, len(bases), 2 * half):
                seqs.append(bases[i:i + half] + mini + bases[i + half:i + 2 * half])
        elif m >= k:
            seqs.append(mini)  # k == m: the bare minimizer is walked as one k-mer (Comparator.cpp:88-90)
        while True:
            e1 = payload.index(b"\n", pos); l1 = payload[pos:e1]; pos = e1 + 1
            e2 = payload.index(b"\n", pos); l2 = payload[pos:e2]; pos = e2 + 1
            if not l1 and not l2:
                break
            seqs.append(l1.decode() + mini + l2.decode())
        buckets[mini] = seqs
        order.append(mini)
    return hdr, buckets, order


def payload_set(payload):
    """{(bucket value, canonical k-mer value)} stored in a sketch payload."""
    hdr, buckets, _ = parse_payload(payload)
    k = hdr["k"]
    s = set()
    for mini, seqs in buckets.items():
        b = val(mini)
        for q in seqs:
            for j in range(len(q) - k + 1):
                s.add((b, canon_val(q[j:j + k])))
    return s


# ------------------------------------------------------------------------------------------------------------
# Second witness for the reference's tie rules.  Written from SubSampler.cpp:81-169 and :357-455
# as DESCRIPTIONS over strings -- occurrences of the winning m-mer, their strands and distances -- not as a
# transcription of the bit-level loop, and not derived from oracle/.
def model_rescan(kmer_str, k, m):
    """what regular_minimizer_pos returns for one k-mer: (canonical value, believed position, is_rev).

    The k-mer is read right to left.  The winner is the canonical m-mer of smallest hash (XXH64 on 8 bytes is a
    bijection, so equal hashes mean equal m-mers); it is first met at its RIGHTMOST occurrence, whose strand
    becomes the winner's strand.  Position: that occurrence's offset -- except that the very first m-mer looked at
    (the rightmost of the k-mer), when it reads reverse, starts at position 0 (:89-93).  Every further occurrence
    to the left ON THE SAME STRAND then pulls the position: a forward winner moves to the leftmost such offset
    (:158-164); a reverse winner compares the position with the DISTANCE FROM THE RIGHT END (the loop index,
    :151-157) and takes the smaller.  Occurrences on the other strand change nothing (:137-149)."""
    km = k - m
    occ = []                                   # (offset, canonical value, hash, reads_reverse), right to left
    for off in range(km, -1, -1):
        sub = kmer_str[off:off + m]
        c = canon_val(sub)
        occ.append((off, c, h64(c), val(sub) != c))
    best_hash = min(o[2] for o in occ)
    hits = [o for o in occ if o[2] == best_hash]          # right to left
    off0, value, _, rev0 = hits[0]
    position = 0 if (off0 == km and rev0) else off0
    for off, _, _, rev in hits[1:]:
        if rev != rev0:
            continue
        if rev0:
            position = min(position, km - off)
        else:
            position = min(position, off)
    return value, position, int(rev0)


def model_scan(seq, k, m, T):
    """the scan loop of one record (SubSampler.cpp:357-455) over a string: ([(start, len, minimizer, rev)] of the
    selected super-k-mers, number of ALL super-k-mers).  State: the current minimizer, its hash, where the scan
    believes it sits and on which strand; `dump` after every rescan."""
    if len(seq) < k:
        return [], 0
    w1 = k - m + 1
    out, total = [], 0
    mini, pos, rev = model_rescan(seq[:k], k, m)
    old_mini, old_rev = mini, rev
    hmin = h64(mini)
    last = 0
    i = 0
    while i + k < len(seq):
        sub = seq[i + w1:i + w1 + m]                      # the m-mer entering on the right
        c = canon_val(sub)
        hc = h64(c)
        dump = False
        if hc < hmin:
            mini, hmin, pos, rev = c, hc, i + w1, int(val(sub) != c)
        elif i >= pos:
            mini, rel, rev = model_rescan(seq[i + 1:i + 1 + k], k, m)
            hmin = h64(mini)
            pos = rel + i + 1
            dump = True
        if old_mini != mini or dump:
            if h64(old_mini) <= T:
                out.append((last, i + k - last, old_mini, old_rev))
            total += 1
            last = i + 1
            old_mini, old_rev = mini, rev
        i += 1
    if len(seq) - last > k - 1:
        if h64(old_mini) <= T:
            out.append((last, len(seq) - last, old_mini, old_rev))
        total += 1
    return out, total


# ------------------------------------------------------------------------------------------------------------------
# second witness for the sketch PAYLOAD (handle_superkmer SubSampler.cpp:243-302, emission :458-504, find_first_kmer
# :604-620, find_next :566-602, reconstruct_superkmer :512-564, strCompressor utils.cpp:48-68): a description over
# strings and Python dicts, written from the reference text -- it shares no code with oracle/ or with the product's
# 2-bit builder, and takes its super-k-mers from model_scan above.
def _compress(bases):
    """strCompressor with the accumulator starting at 0: first byte len % 4, then 4 bases per byte, first
    base in the top bits; a partial last byte carries one extra shift"""
    if not bases:
        return b""
    out = bytearray([len(bases) % 4])
    c = 0
    for i, ch in enumerate(bases):
        c = (c + ((ord(ch) >> 1) & 3)) & 0xff
        if (i + 1) % 4 == 0:
            out.append(c)
            c = 0
        c = (c << 2) & 0xff
    if len(bases) % 4:
        out.append(c)
    return bytes(out)


def model_payload(records, k, m, T, rate, abundance=1):
    """the bytes parse_fasta_test hands to its gzip writer for these (already cleaned) records"""
    buckets = {}                 # minimizer value -> {k-mer string: [count (uint8), position of the minimizer, seen]}, insertion order
    selected = 0
    for seq in records:
        for start, ln, mini, rev in model_scan(seq, k, m, T)[0]:
            s = seq[start:start + ln]
            if rev:
                s = rc_str(s)
            selected += len(s) - k + 1
            mstr = to_str(mini, m)
            d = buckets.setdefault(mini, {})
            for i in range(len(s) - k + 1):
                km = s[i:i + k]
                if km in d:
                    d[km][0] = (d[km][0] + 1) & 0xff
                else:
                    d[km] = [1, km.find(mstr) & 0xff, False]
    out = bytearray(("%d %d %d %f\n" % (k - 1 + (k - m + 1), m, selected, rate)).encode())

    def usable(e):
        return (not e[2]) and e[0] >= abundance

    for mini in sorted(buckets):
        d = buckets[mini]
        mstr = to_str(mini, m)
        out += mstr.encode()

        def take(cand):
            e = d.get(cand)
            if e is not None and usable(e):
                e[2] = True
                return True
            return False

        def find_next(cur, left):
            for nuc in "ATCG":                                   # the order the reference tries them in (:568)
                cand = nuc + cur[:-1] if left else cur[1:] + nuc
                if take(cand):
                    return cand
            return cur

        max_bases, text = "", ""
        while True:
            first = next((km for km, e in d.items() if usable(e)), None)
            if first is None:
                break
            d[first][2] = True
            sk = first
            n_left, n_right = (k - m) - d[first][1], d[first][1]
            cur = first
            while len(sk) != 2 * k - m:
                if n_left != 0:
                    nx = find_next(cur, True)
                    n_left -= 1
                    if nx != cur:
                        sk = nx[0] + sk
                    else:
                        n_left = 0
                    cur = first if n_left == 0 else nx
                elif n_right != 0:
                    nx = find_next(cur, False)
                    n_right -= 1
                    if nx == cur:
                        break
                    sk += nx[-1]
                    cur = nx
                else:
                    break
            if len(sk) == 2 * k - m:
                max_bases += sk[:k - m] + sk[k:k + (k - m)]
            else:
                p = sk.find(mstr)
                text += sk[:p] + "\n" + sk[p + m:] + "\n"
        blob = _compress(max_bases)
        out += len(blob).to_bytes(4, "little") + blob + text.encode() + b"\n\n"
    return bytes(out)

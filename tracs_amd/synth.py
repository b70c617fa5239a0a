"""Seeded synthetic alignments, dates and count tables.

The mutation model follows the idea of the reference's simulator (scripts/tracs-sim.py:10-46:
substitute a different base at chosen positions); the code is ours.  A two-level star phylogeny
gives outbreak-like SNP distances: lineage founders differ from the ancestor at Bernoulli(mu_lineage)
sites, samples from their founder at Bernoulli(mu_sample) sites; Bernoulli(p_n) sites become 'N'.
"""
import numpy as np

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_IUPAC_PARTIAL = np.frombuffer(b"MRWSYKVHDB", dtype=np.uint8)
_OTHER = np.frombuffer(b"N-?Xn.*", dtype=np.uint8)


def _mutate(idx, rng, mu):
    """idx: int8[..., L] base indices 0..3 -> copy with a different base at Bernoulli(mu) sites."""
    out = idx.copy()
    L = idx.shape[-1]
    k = rng.binomial(L, mu) if mu > 0 else 0
    if k:
        pos = rng.choice(L, size=k, replace=False)
        out[pos] = (out[pos] + rng.integers(1, 4, size=k)) & 3
    return out


def alignment(n, L, seed, mu_lineage=1e-4, mu_sample=1e-5, n_lineages=None, p_n=0.01, p_partial=0.0,
              p_lower=0.0, p_other=0.0):
    """-> uint8[n, L] ASCII.  p_partial: IUPAC two/three-base codes; p_lower: lower-case letters;
    p_other: '-', '?', 'X', '.', '*' and friends (all of which the reference treats as N)."""
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, size=L, dtype=np.int8)
    n_lineages = max(1, n // 16) if n_lineages is None else n_lineages
    founders = [_mutate(anc, rng, mu_lineage) for _ in range(n_lineages)]
    out = np.empty((n, L), dtype=np.uint8)
    for s in range(n):
        idx = _mutate(founders[s % n_lineages], rng, mu_sample)
        row = _BASES[idx]
        if p_n > 0:
            row[rng.random(L) < p_n] = ord("N")
        if p_partial > 0:
            m = rng.random(L) < p_partial
            row[m] = _IUPAC_PARTIAL[rng.integers(0, len(_IUPAC_PARTIAL), size=int(m.sum()))]
        if p_other > 0:
            m = rng.random(L) < p_other
            row[m] = _OTHER[rng.integers(0, len(_OTHER), size=int(m.sum()))]
        if p_lower > 0:
            m = (rng.random(L) < p_lower) & (row >= 65) & (row <= 90)
            row[m] = row[m] + 32
        out[s] = row
    return out


def write_fasta(path, seqs, names=None, width=0, gz=False):
    """width = 0: one line per record; else wrapped."""
    import gzip
    names = names or ["s%d" % i for i in range(seqs.shape[0])]
    op = gzip.open if gz else open
    with op(path, "wb") as fh:
        for nm, row in zip(names, seqs):
            fh.write(b">" + nm.encode() + b"\n")
            b = row.tobytes()
            if width:
                for o in range(0, len(b), width):
                    fh.write(b[o:o + width] + b"\n")
            else:
                fh.write(b + b"\n")
    return names


def dates(n, seed, span_days=730, start="2020-01-01"):
    """-> (iso strings, integer days since 1970-01-01), uniform in [start, start+span_days)."""
    from datetime import date, timedelta
    rng = np.random.default_rng(seed)
    d0 = date.fromisoformat(start)
    off = rng.integers(0, span_days, size=n)
    ds = [d0 + timedelta(days=int(o)) for o in off]
    epoch = date(1970, 1, 1)
    return [d.isoformat() for d in ds], np.array([(d - epoch).days for d in ds], dtype=np.int32)


def allele_counts(L, seed, depth=30, eps=0.01, p_two=0.01):
    """-> uint16[L, 4] pileup counts: Multinomial(Poisson(depth), (1-eps, eps/3, ...)) on a random
    major allele, with a fraction p_two of two-allele sites at 0.7/0.3."""
    rng = np.random.default_rng(seed)
    major = rng.integers(0, 4, size=L)
    dep = rng.poisson(depth, size=L)
    p = np.full((L, 4), eps / 3)
    p[np.arange(L), major] = 1 - eps
    two = rng.random(L) < p_two
    minor = (major + rng.integers(1, 4, size=L)) & 3
    p[two] = eps / 2
    p[two, major[two]] = 0.7 * (1 - eps)
    p[two, minor[two]] = 0.3 * (1 - eps)
    p /= p.sum(1, keepdims=True)
    # multinomial per row via sequential binomials (vectorised)
    out = np.zeros((L, 4), dtype=np.int64)
    rem = dep.copy()
    prem = np.ones(L)
    for k in range(3):
        q = np.clip(p[:, k] / prem, 0, 1)
        out[:, k] = rng.binomial(rem, q)
        rem -= out[:, k]
        prem -= p[:, k]
    out[:, 3] = rem
    return out.astype(np.uint16)


# ---- on-device generation for the benchmark (torch) -------------------------------------
def coverage_runs(L, seed, p_n=0.01, mean_len=500, frac_lo=0.005, frac_hi=0.30):
    """Coverage gaps as real consensus alignments have them: RUNS of consecutive sites (geometric lengths, mean `mean_len`), each
    lost by a random fraction of the samples (uniform in [frac_lo, frac_hi]) -- runs are drawn until a sample is N at p_n of the
    sites on average.  -> (start, end, fraction) arrays, in draw order; sample s loses run r iff member(s)[r] (below)."""
    rng = np.random.default_rng([int(seed), 7919])
    starts, ends, fracs, covered = [], [], [], 0.0
    while covered < p_n * L:
        ln = int(min(L, rng.geometric(1.0 / mean_len)))
        st = int(rng.integers(0, L - ln + 1))
        fr = float(rng.uniform(frac_lo, frac_hi))
        starts.append(st); ends.append(st + ln); fracs.append(fr)
        covered += ln * fr
    return np.array(starts, dtype=np.int64), np.array(ends, dtype=np.int64), np.array(fracs)


def run_members(seed, s, fracs):
    """which runs sample s loses (deterministic in seed and s, whatever the batch it is generated in)"""
    return np.random.default_rng([int(seed), 104729, int(s)]).random(len(fracs)) < fracs


def generate_device(n, L, seed, emit, mu_lineage=1e-5, mu_sample=1e-6, n_lineages=None, p_n=0.01, batch=32,
                    limit=None, p_partial=0.0, n_every=1, runs=None, gaps=None):
    """The same two-level model generated on the GPU in batches of `batch` samples; every batch
    (uint8 [cnt, L] ASCII on the device) is handed to emit(rows, first).  Deterministic in `seed`;
    `limit` stops after the first `limit` samples (same values as a full run).  n_every = k: only every k-th sample carries
    N, at k p_n sites (the same amount of N, concentrated in 1 / k of the samples).  runs = dict(p_n=, mean_len=, frac_lo=,
    frac_hi=): N in runs of consecutive sites shared by subsets of the samples (coverage_runs).  gaps = dict(frac=, mean_len=): every
    sample is N over `frac` of the sites in runs of geometric length (mean mean_len) whose boundaries are its own -- what `tracs
    align` writes where a sample's coverage is below its thresholds (tracs/align.py:599-613).  Setup code, untimed."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    host_rng = np.random.default_rng(int(seed))
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    plut = torch.tensor(list(b"MRWSYKVHDB"), dtype=torch.uint8, device=dev)
    anc = torch.randint(0, 4, (L,), generator=g, device=dev, dtype=torch.int8)
    n_lineages = max(1, n // 16) if n_lineages is None else n_lineages

    def mutate(base, mu):
        k = int(host_rng.poisson(L * mu)) if mu > 0 else 0
        out = base.clone()
        if k:
            # distinct positions: an indexed store with a repeated index keeps whichever write lands last (not reproducible)
            pos = torch.unique(torch.randint(0, L, (k,), generator=g, device=dev))
            out[pos] = (out[pos] + torch.randint(1, 4, (k,), generator=g, device=dev, dtype=torch.int8)[:pos.numel()]) & 3
        return out
    founders = [mutate(anc, mu_lineage) for _ in range(n_lineages)]
    if runs:                                  # N in runs of sites shared by subsets of the samples (coverage_runs) -- beside p_n's iid N
        r_start, r_end, r_frac = coverage_runs(L, seed, **runs)
        r_start_d, r_end_d = torch.from_numpy(r_start).to(dev), torch.from_numpy(r_end).to(dev)
    stop = n if limit is None else min(n, limit)
    for s0 in range(0, stop, batch):
        cnt = min(batch, n - s0)
        rows = torch.empty((cnt, L), dtype=torch.uint8, device=dev)
        for b in range(cnt):
            idx = mutate(founders[(s0 + b) % n_lineages], mu_sample)
            rows[b] = lut[idx.long()]
            if p_n > 0 and (s0 + b) % n_every == 0:
                m = torch.rand(L, generator=g, device=dev) < min(1.0, p_n * n_every)
                rows[b][m] = ord("N")
            if runs:
                mem = torch.from_numpy(run_members(seed, s0 + b, r_frac)).to(dev)
                edge = torch.zeros(L + 1, dtype=torch.int32, device=dev)
                one = torch.ones(int(mem.sum().item()), dtype=torch.int32, device=dev)
                edge.index_add_(0, r_start_d[mem], one)
                edge.index_add_(0, r_end_d[mem], -one)
                rows[b][torch.cumsum(edge, 0)[:L] > 0] = ord("N")
            if p_partial > 0:       # two/three-allele IUPAC codes at uniformly random sites (SURVEY.md 8d, config 4 mix)
                m = torch.rand(L, generator=g, device=dev) < p_partial
                k = int(m.sum().item())
                rows[b][m] = plut[torch.randint(0, 10, (k,), generator=g, device=dev)]
            if gaps:                # per-sample coverage gaps (last: a masked site is N whatever was called there): k runs cover 1 - exp(-k mean_len / L) of the sites
                k = max(1, int(round(-np.log(1.0 - gaps["frac"]) * L / gaps["mean_len"])))
                st = torch.randint(0, L, (k,), generator=g, device=dev)
                u = torch.rand(k, generator=g, device=dev, dtype=torch.float64).clamp_(1e-12, 1.0)
                ln = (torch.log(u) / np.log1p(-1.0 / gaps["mean_len"])).floor().long() + 1        # geometric, mean mean_len
                en = torch.minimum(st + ln, torch.tensor(L, device=dev))
                edge = torch.zeros(L + 1, dtype=torch.int32, device=dev)
                one = torch.ones(k, dtype=torch.int32, device=dev)
                edge.index_add_(0, st, one)
                edge.index_add_(0, en, -one)
                rows[b][torch.cumsum(edge, 0)[:L] > 0] = ord("N")
        emit(rows, s0)
    torch.cuda.synchronize()


def pack_synthetic_device(aln, seed, **kw):
    """Fill a tracs_amd.device.Alignment from generate_device (the ASCII never exists on the host)."""
    generate_device(aln.n, aln.L, seed, lambda rows, first: aln.pack(rows, first=first), **kw)


def first_samples_host(n, L, seed, m, **kw):
    """The first m samples of the same synthetic alignment, as a host uint8 [m, L] array."""
    chunks = []
    generate_device(n, L, seed, lambda rows, first: chunks.append(rows.cpu().numpy()), limit=m, **kw)
    return np.concatenate(chunks, axis=0)[:m]

"""Chain sharding across the GPUs of a node and the cross-chain diagnostics reduction.

The reference's only parallelism is one task per chain (tbb::parallel_for over chains,
mcmc.h:1494-1508, 1534-1548); chains share nothing and chain i always draws from
pcg32(0, i+1) (mcmc.h:999,1527).  So the path shards by chain-id range with NO collective
on the sampling path: rank g of G owns chains [g*N/G, (g+1)*N/G) and results per chain are
independent of G.  The only exchange is the small additive diagnostics vector (acceptance,
R-hat and ESS ingredients), all-reduced over RCCL (`nccl` backend) or gloo in CPU tests.
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous chain range [lo, hi) of `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    lo = rank * n_total // world
    hi = (rank + 1) * n_total // world
    return lo, hi


def shard_chain_ids(n_total, rank, world, first_chain_id=1):
    """Chain ids (= pcg32 stream ids, mcmc.h:1527: i+1) of this rank's chains."""
    lo, hi = shard_range(n_total, rank, world)
    return np.arange(first_chain_id + lo, first_chain_id + hi, dtype=np.uint64)


def allreduce_sum(vec, group=None):
    """Sum an additive float64 vector over all ranks (identity if torch.distributed is not
    initialised).  Uses the backend's device: cuda for nccl(=RCCL), cpu for gloo."""
    import torch
    import torch.distributed as dist
    v = np.ascontiguousarray(vec, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return v.copy()
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.from_numpy(v.copy()).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def moment_sums(samples, max_lag=0):
    """Additive ingredients of R-hat / ESS for this shard.

    samples: [n_chains, n_saved, k] post-burn-in draws of k scalars.
    Returns a flat float64 vector: [n_chains, n_saved, then per scalar: sum of chain means,
    sum of squared chain means, sum of chain variances (ddof=1), sum over chains of the
    lag-0..max_lag autocovariances]."""
    s = np.asarray(samples, dtype=np.float64)
    m, n, k = s.shape
    mu = s.mean(axis=1)
    var = s.var(axis=1, ddof=1) if n > 1 else np.zeros((m, k))
    d = s - mu[:, None, :]
    acov = np.zeros((max_lag + 1, k))
    for lag in range(max_lag + 1):
        acov[lag] = (d[:, :n - lag, :] * d[:, lag:, :]).sum(axis=1).sum(axis=0) / n
    parts = [np.array([m, n], dtype=np.float64), mu.sum(0), (mu ** 2).sum(0), var.sum(0), acov.ravel()]
    return np.concatenate(parts)


def summarize(sums, k, max_lag=0):
    """R-hat (Gelman-Rubin) and ESS (Geyer initial positive sequence on the pooled
    autocovariance) from all-reduced moment_sums.  Extension: the reference only prints a
    per-chain acceptance rate (mcmc.h:1136)."""
    m, n_times_world = sums[0], sums[1]
    # n_saved is identical on every rank; the all-reduce multiplied it by the world size
    sum_mu, sum_mu2, sum_var = sums[2:2 + k], sums[2 + k:2 + 2 * k], sums[2 + 2 * k:2 + 3 * k]
    acov = sums[2 + 3 * k:].reshape(max_lag + 1, k) / m
    return dict(m=m, sum_mu=sum_mu, sum_mu2=sum_mu2, sum_var=sum_var, acov=acov, n_field=n_times_world)


def rhat_ess(sums, k, n_saved, max_lag=0):
    d = summarize(sums, k, max_lag)
    m, n = d["m"], float(n_saved)
    W = d["sum_var"] / m
    grand = d["sum_mu"] / m
    B_over_n = (d["sum_mu2"] - m * grand ** 2) / max(m - 1, 1)
    var_plus = (n - 1) / n * W + B_over_n
    with np.errstate(divide="ignore", invalid="ignore"):
        rhat = np.sqrt(var_plus / W)
    ess = np.full(k, m * n)
    if max_lag > 0:
        rho = d["acov"] / np.where(d["acov"][0] > 0, d["acov"][0], 1.0)
        for j in range(k):
            tau = 1.0
            t = 1
            while t + 1 <= max_lag:
                pair = rho[t, j] + rho[t + 1, j]
                if pair <= 0:
                    break
                tau += 2 * pair
                t += 2
            ess[j] = m * n / tau
    return rhat, ess


def acceptance_from_diagnostics(diag_sum, n_chains_total, steps_per_chain):
    """diag vector of mcmc_hip_diagnostics (additive over shards) -> acceptance rate."""
    return diag_sum[0] / (n_chains_total * max(steps_per_chain, 1))


def ess_geyer(series):
    """Effective sample size of every chain by Geyer's initial positive sequence.

    series: [n_chains, n] draws of one scalar.  Returns ess [n_chains] (n / tau, with
    tau = -1 + 2 * sum of the leading positive pair sums rho[2t] + rho[2t+1]); a chain that
    never moved has ESS 0.  Extension: the reference reports no ESS."""
    s = np.asarray(series, dtype=np.float64)
    m, n = s.shape
    d = s - s.mean(axis=1, keepdims=True)
    f = np.fft.rfft(d, 2 * n, axis=1)
    acov = np.fft.irfft(f * np.conj(f), 2 * n, axis=1)[:, :n] / n
    var = acov[:, 0]
    ok = var > 0
    rho = acov / np.where(ok, var, 1.0)[:, None]
    npair = n // 2
    pairs = rho[:, 0:2 * npair:2] + rho[:, 1:2 * npair:2]
    positive = np.cumprod(pairs > 0, axis=1)               # 1 until the first non-positive pair
    tau = -1.0 + 2.0 * (pairs * positive).sum(axis=1)
    return np.where(ok, n / np.maximum(tau, 1.0 / n), 0.0)

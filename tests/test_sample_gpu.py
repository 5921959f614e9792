"""On-device temperature / top-k / top-p sampling (MI355X) against the ORACLE's restatement of api.cpp:126-216
(oracle/tl_oracle.c tlo_sample_candidates / tlo_discrete_draw, written independently of the product's host port in
host/tl_session.cpp, which tests/test_host_cpu.py checks against the same oracle). Parity unpinned: api.cpp cannot be
compiled here (Boost) and the reference holds no sampling vectors, so the oracle is a reading of the source.

Bars: the candidate ids in draw order are identical wherever neighbouring probabilities differ by more than one FP32
ulp-scale gap (libm vs device exp may round a probability differently in its last bit), renormalised probabilities
within 2e-6 relative, and the drawn id is the inverse-CDF pick for the given uniform variate. Whole-model sampled
generation: reproducible per seed, equal to greedy at temperature < 0.05, and its first-token histogram over seeds
follows the candidate probabilities."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E(pkg):
    from tinyllama_cpp_amd import engine
    engine.init(0)
    return engine


@pytest.fixture(scope="module")
def H(pkg):
    from tinyllama_cpp_amd import hostapi
    hostapi.lib()
    return hostapi


def _same_candidates(ids, probs, wi, wp):
    assert len(ids) == len(wi)
    assert np.allclose(probs, wp, rtol=2e-6, atol=0)
    diff = np.nonzero(ids != wi)[0]
    for i in diff:  # only allowed inside a run of (nearly) equal probabilities
        j = int(np.nonzero(wi == ids[i])[0][0])
        assert abs(wp[j] - wp[i]) <= 4e-7 * wp[i], (i, j)


@pytest.mark.parametrize("V,T,k,pp", [(1000, 0.8, 40, 0.9), (128256, 0.7, 40, 0.9), (32000, 1.5, 1024, 0.99),
                                      (257, 1.0, 0, 0.5), (64, 0.1, 5, 0.999), (5000, 1.0, 200, 1.0), (100, 0.7, 50, 0.0)])
def test_device_sampler_matches_oracle(E, oracle, V, T, k, pp):
    rng = np.random.default_rng(V + k)
    logits = (rng.standard_normal(V) * 3).astype(np.float32)
    logits[3] = logits[7] = logits[11]  # equal probabilities: the higher id sorts first
    wi, wp = oracle.sample_candidates(logits, T, k, pp)
    for u in (0.0, 0.25, 0.5, 0.9, 0.999999):
        got, ids, probs = E.sample_top_k_top_p(logits, T, k, pp, u)
        _same_candidates(ids, probs, wi, wp)
        # the draw of std::discrete_distribution (oracle: libstdc++'s lower_bound over double partial sums) on the
        # device's own candidate list
        assert got == int(ids[oracle.discrete_draw(probs, u)]), (u, got)


def test_device_sampler_edge_cases(E):
    from tinyllama_cpp_amd.capi import TlError
    rng = np.random.default_rng(1)
    logits = (rng.standard_normal(3000) * 2).astype(np.float32)
    assert E.sample_top_k_top_p(logits, 0.01, 40, 0.9, 0.7)[0] == int(np.argmax(logits))  # temperature < 0.05
    flat = np.zeros(500, np.float32)  # all equal: descending id order, uniform probabilities
    got, ids, probs = E.sample_top_k_top_p(flat, 1.0, 10, 1.0, 0.0)
    assert ids.tolist() == list(range(499, 489, -1)) and np.allclose(probs, 0.1) and got == 499
    spike = np.full(4000, -50.0, np.float32)
    spike[1234] = 20.0  # everything else underflows to probability 0
    got, ids, probs = E.sample_top_k_top_p(spike, 1.0, 40, 0.9, 0.99)
    assert got == 1234 and ids[0] == 1234 and len(ids) == 1
    one = E.sample_top_k_top_p(logits, 1.0, 1, 0.9, 0.5)
    assert one[0] == int(np.argmax(logits)) and len(one[1]) == 1
    with pytest.raises(TlError):
        E.sample_top_k_top_p(logits, 1.0, 0, 0.9, 0.5)  # "no top-k" over 3000 ids: host path only
    with pytest.raises(TlError):
        E.sample_top_k_top_p(logits, 1.0, 40, 0.9, 1.0)  # u outside [0,1)


def test_model_generate_sampled(E, oracle, pkg):
    cfg, tensors = pkg.synth.build_model("tiny-gqa", "q4_k_m", seed=5)
    m = E.Model(cfg).load(tensors)
    prompt = [3, 99, 250, 7]
    def run(**kw):
        m.reset_kv()
        for pos, t in enumerate(prompt[:-1]):
            m.decode(t, pos, want_logits=False)
        return m.generate_sampled(prompt[-1], len(prompt) - 1, 16, **kw)
    m.reset_kv()
    for pos, t in enumerate(prompt[:-1]):
        m.decode(t, pos, want_logits=False)
    greedy = m.generate_greedy(prompt[-1], len(prompt) - 1, 16, 1.3)
    assert np.array_equal(run(penalty=1.3, temperature=0.0, top_k=40, top_p=0.9, seed=1), greedy)
    a = run(penalty=1.3, temperature=1.0, top_k=50, top_p=0.95, seed=11)
    b = run(penalty=1.3, temperature=1.0, top_k=50, top_p=0.95, seed=11)
    c = run(penalty=1.3, temperature=1.0, top_k=50, top_p=0.95, seed=12)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert ((a >= 0) & (a < cfg["vocab"])).all()
    # first sampled token over many seeds follows the host-computed candidate distribution of the same logits
    m.reset_kv()
    for pos, t in enumerate(prompt[:-1]):
        m.decode(t, pos, want_logits=False)
    logits, _ = m.decode(prompt[-1], len(prompt) - 1)
    wi, wp = oracle.sample_candidates(logits, 1.2, 8, 1.0)
    counts = dict.fromkeys(wi.tolist(), 0)
    N = 400
    for seed in range(N):
        first = int(run(penalty=1.0, temperature=1.2, top_k=8, top_p=1.0, seed=1000 + seed)[0])
        assert first in counts
        counts[first] += 1
    chi2 = sum((counts[i] - N * p) ** 2 / (N * p) for i, p in zip(wi.tolist(), wp.tolist()) if N * p > 5)
    assert chi2 < 30.0, (chi2, counts)  # 7 degrees of freedom: P(chi2 > 30) ~ 1e-4
    assert len(m.generate_sampled(1, 0, 4, temperature=1.0, top_k=0, top_p=0.9)) == 4  # "no top-k": vocab 512 <= 1024
    m.free()

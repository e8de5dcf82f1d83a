"""bdpt_query_rays_noalloc (the entry point for callers that must not allocate device memory) against
bdpt_query_rays on the same rays: ids, hit points and normals bit for bit, on two calls in a row, the second
smaller than the first, so that what the first left in the shared staging array would show."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rays(rng, n):
    o = rng.uniform(-2.9, 2.9, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    org = np.tile(np.array([4, 0], np.int32), (n, 1))
    org[: n // 4] = [0, 0]
    org[n // 4: n // 2] = [2, 3]
    return np.concatenate([o, d], 1), org


def _same(a, b):
    ids_a, pos_a, nrm_a = a
    ids_b, pos_b, nrm_b = b
    assert np.array_equal(ids_a, ids_b)
    hit = ids_a[:, 0] >= 0
    assert hit.any() and (~hit).any() or hit.all()
    assert np.array_equal(pos_a[hit].view(np.uint32), pos_b[hit].view(np.uint32))
    assert np.array_equal(nrm_a[hit].view(np.uint32), nrm_b[hit].view(np.uint32))


def test_noalloc_ray_query_equals_the_allocating_one_on_two_calls_in_a_row(bdpt):
    rng = np.random.default_rng(2025)
    big, big_org = _rays(rng, 5000)
    small, small_org = _rays(rng, 700)
    with bdpt.Context(8, 8) as ctx:
        ctx.upload_scene(bdpt.Scene.soup(300))
        want_big = ctx.query_rays(big, big_org)
        want_small = ctx.query_rays(small, small_org)
        _same(ctx.query_rays(big, big_org, noalloc=True), want_big)
        _same(ctx.query_rays(small, small_org, noalloc=True), want_small)
        # a second context on the same GPU shares the array; used one after the other that is fine
        with bdpt.Context(16, 16) as other:
            other.upload_scene(bdpt.Scene.cornell())
            want = other.query_rays(small, small_org)
            _same(other.query_rays(small, small_org, noalloc=True), want)
        _same(ctx.query_rays(big, big_org, noalloc=True), want_big)
        # the capacity is part of the contract
        n_max = bdpt.QUERY_NOALLOC_MAX_RAYS
        full, full_org = _rays(rng, n_max)
        _same(ctx.query_rays(full, full_org, noalloc=True), ctx.query_rays(full, full_org))
        over, over_org = _rays(rng, n_max + 1)
        with pytest.raises(bdpt.BdptError, match="at most"):
            ctx.query_rays(over, over_org, noalloc=True)
        assert ctx.query_rays(over[:0], over_org[:0], noalloc=True)[0].shape == (0, 2)

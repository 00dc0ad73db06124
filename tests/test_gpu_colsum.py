import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum_periodic(dtype):
    from cddmsl_amd import hip
    g = torch.Generator().manual_seed(0)
    x = torch.randn(50 * 37 + 13, 96, generator=g).to(dtype)
    ref = x.float().sum(0)
    out = hip.colsum(x.cuda())
    assert (out.cpu() - ref).abs().max() < 1e-3 * max(1.0, float(ref.abs().max()))
    x2 = x[: 50 * 37]
    ref2 = x2.float().view(37, 50, 96).sum(0)
    out2 = hip.colsum(x2.cuda(), period=50)
    assert (out2.cpu() - ref2).abs().max() < 1e-3 * max(1.0, float(ref2.abs().max()))

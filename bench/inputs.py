"""Device-side random inputs for the bench tools and the GPU tests (torch's generator: inputs that do NOT come
from the SplitMix64 stream the library itself can produce)."""


def random_elements(torch, n, seed):
    """n random elements of the src/bn256.rs field as (n, 4) int64 limbs on the current device:
    three uniform 64-bit limbs and a top limb below floor(p / 2^224) * 2^32, hence value < p."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    out = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    out[:, :3] = torch.randint(-2**63, 2**63 - 1, (n, 3), dtype=torch.int64, device="cuda", generator=g)
    out[:, 3] = torch.randint(0, 0x73EDA753 << 32, (n,), dtype=torch.int64, device="cuda", generator=g)
    return out

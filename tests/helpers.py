"""Shared comparison helpers for the parity tests."""
import torch


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def max_abs(a: torch.Tensor, b: torch.Tensor) -> float:
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def assert_close(a, b, rel, what="", abs_floor=0.0):
    err = rel_l2(a, b)
    if err > rel and max_abs(a, b) > abs_floor:
        raise AssertionError(f"{what}: rel-L2 {err:.3e} > {rel:.1e} (max-abs {max_abs(a, b):.3e})")
    return err

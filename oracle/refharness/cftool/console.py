"""cftool.console shell (rich-styled logging in the real package)."""
from typing import Any


def log(msg: str, *a: Any, **k: Any) -> None:
    print(msg)


def debug(msg: str, *a: Any, **k: Any) -> None:
    print(msg)


def warn(msg: str, *a: Any, **k: Any) -> None:
    print(f"[warn] {msg}")


def error(msg: str, *a: Any, **k: Any) -> None:
    print(f"[error] {msg}")


def rule(*a: Any, **k: Any) -> None:
    pass


def print(*a: Any, **k: Any) -> None:  # noqa: A001
    import builtins

    builtins.print(*a, **k)

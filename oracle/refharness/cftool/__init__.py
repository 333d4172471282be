"""Minimal stand-in for the un-vendored third-party dependency `carefree-toolkit`
(`cftool`, pinned only as `>=0.3.12` by the reference's setup.py:45).

TEST INFRASTRUCTURE ONLY.  It exists so that the reference's own hot-path modules can be
imported from /root/reference (read-only) inside the build container, to (a) validate the CPU
restatement in `oracle/vit_oracle.py` and (b) generate the golden fixtures in `tests/golden/`.
Nothing in the product path (`carefree-learn_amd/`) imports this.

Behaviour is restated from the reference's call sites; the real package is not
available offline.
"""

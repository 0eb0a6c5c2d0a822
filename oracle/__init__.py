"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU restatement (pure torch-CPU / numpy) of the reference's algorithm for the DeepLIO
training hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package, and only as the checker / the timed CPU baseline.  Nothing under
deeplio_amd/ imports it; the product path fails loudly without the HIP extension.

Pinning: every module here is checked against golden vectors captured by importing the
reference itself (tests/golden/make_golden.py, run in the build container where
/root/reference exists; fixtures committed under tests/golden/).  Third-party arithmetic
that is absent from the reference tree (liegroups.torch.SO3, torchvision BasicBlock; both
version-unpinned upstream) is restated from the published algorithm and cross-checked
against the reference's in-tree deeplio/common/spatial.py equivalents -- see se3.py.
"""

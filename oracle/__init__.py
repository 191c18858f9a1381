"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

A CPU restatement of the reference algorithm for the hot path, written with
plain torch-CPU functional ops and numpy.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this package, and only as the checker / reported CPU
baseline -- never on the product path (flood_uav_video_segmentation_amd/* must not import it).

Pinning: the reference has no tests, fixtures or golden vectors of its own, so
the oracle is pinned against outputs of the reference itself, produced in the build container by
importing /root/reference (tests/golden/gen_goldens.py) and committed as fixtures under
tests/golden/.  DeepLabv3 is the exception: its arithmetic lives in torchvision (pinned 0.12.0 in
the reference's Pipfile.lock; model fetched by torch.hub at tag v0.10.0, model/deeplabv3.py:15),
which is absent offline -> oracle/deeplab_oracle.py restates the public architecture and is
"parity unpinned".
"""

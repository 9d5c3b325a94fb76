"""CPU oracle for the SiMHand contrastive pre-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``simhand_amd/`` may import this
package; the only legal importers are ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` (where it is the checker / the
timed CPU baseline, never the product).

Parity pinning: ``oracle/make_golden.py`` executes the reference's own Python
(``/root/reference/src/models/utils.py`` and the ``HandCLR_W`` / ``PeCLR_W`` /
``SimCLR`` step classes) in the build container through the stub importer in
``oracle/ref_import.py`` and stores inputs + outputs under ``tests/golden``.
``tests/test_oracle_golden.py`` checks every function here against those
vectors.  Pieces whose source is NOT under /root/reference (torchvision 0.13.1
ResNet, pl_bolts 0.2.2 LARSWrapper / LinearWarmupCosineAnnealingLR) are
restated from their published definitions and are marked "parity unpinned"
where they are defined.
"""

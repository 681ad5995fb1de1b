"""Minimal stand-in for the parts of timm 0.3.2 the DiG reference imports.
Test infrastructure only (used by oracle/ref_harness to import /root/reference in the build container)."""

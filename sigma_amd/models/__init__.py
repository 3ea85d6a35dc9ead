"""Host-side mirror of the reference's ``models`` package for the Sigma (VMamba) path only.

``sigma_amd.models.builder.EncoderDecoder`` is drop-in boundary #1; see
INTEGRATION.md for how the reference's train.py / eval.py pick it up unchanged.
"""

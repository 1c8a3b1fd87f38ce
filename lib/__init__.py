"""Drop-in package with the reference's import paths (``lib.models.builder.build_model``,
``lib.models.MicKey.compute_pose.MickeyRelativePose``, ``lib.utils.data.data_to_model_device``), backed
by mickey_amd.  The reference's callers (submission.py:13,89; demo_inference.py:3,91) import exactly
these; everything else they import (datasets, config, visualisation, metrics) stays the reference's own
code -- see INTEGRATION.md."""

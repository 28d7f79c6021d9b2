"""MI355X-native bundle-adjustment engine for the Optimizer/ hot path of large-scale-sfm.

The package directory is named ``large-scale-sfm_amd`` (hyphens): import it with
``importlib.import_module("large-scale-sfm_amd")``.
"""

"""treegp_amd -- MI355X-native GP interpolation hot path with treegp's Python API.

Mirrors ``treegp/__init__.py:7-36`` of the reference for the names on the hot path.
"""
__version__ = "0.1.0"

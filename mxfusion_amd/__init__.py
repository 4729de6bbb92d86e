"""mxfusion_amd -- MI355X-native Gaussian-process + SVI hot path behind MXFusion's API."""
__version__ = '0.1.0'

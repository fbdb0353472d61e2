from .fourier_features import FourierFeatures  # noqa: F401

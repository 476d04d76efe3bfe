from .tree_gps import BARKModel, forest_predict, mixture_of_gaussians_as_normal  # noqa: F401

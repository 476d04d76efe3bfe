from .tree_gps import forest_predict, mixture_of_gaussians_as_normal  # noqa: F401

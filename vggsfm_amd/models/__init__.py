from .triangulator import Triangulator, find_best_initial_pair  # noqa: F401

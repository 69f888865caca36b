"""Drop-in counterpart of the sampler half of the reference's cvrp_nls/ directory."""

from stemseg_amd.modeling.embedding_utils import (add_spatiotemporal_offset, creat_spatiotemporal_grid,  # noqa: F401
                                                  get_nb_embedding_dims, get_nb_free_dims)

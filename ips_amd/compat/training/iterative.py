from ips_amd.training.iterative import (compute_loss, evaluate, fill_batch, init_batch,  # noqa: F401
                                        shrink_batch, train_one_epoch)

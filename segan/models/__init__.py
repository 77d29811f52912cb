from segan_pytorch_amd.models import *  # noqa: F401,F403
from segan_pytorch_amd.models import (Discriminator, GConv1DBlock, GDeconv1DBlock, Generator,  # noqa
                                      GSkip, Model, Saver, SEGAN, WSEGAN, weights_init,
                                      wsegan_weights_init)

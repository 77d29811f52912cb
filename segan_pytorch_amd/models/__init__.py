from .core import Model, Saver
from .modules import GConv1DBlock, GDeconv1DBlock, build_norm_layer
from .generator import Generator, GSkip
from .discriminator import Discriminator
from .model import SEGAN, WSEGAN, weights_init, wsegan_weights_init

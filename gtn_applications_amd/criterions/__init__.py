"""Drop-in counterparts of /root/reference/criterions/*: same class / function names, argument
meaning and error behaviour; the arithmetic runs on the MI355X kernels of libwfl.so."""
from . import asg, ctc, stc, transducer  # noqa: F401

from .loadProtocol import loadProtocol  # noqa: F401
from .protocolBase import PacketEndDetect, PacketLenEndianness  # noqa: F401

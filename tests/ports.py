"""A free TCP port on the loopback interface for a process-group rendezvous (test infrastructure)."""
import socket


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]

"""a free TCP port on the loopback interface for the rendezvous of the multi-process tests"""
import socket


def free_port():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]

"""Authoring-container-only helpers that import the reference from /root/reference."""

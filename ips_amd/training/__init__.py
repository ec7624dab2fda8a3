"""Caller-side training / evaluation loops (mirror of the reference's ``training`` package)."""

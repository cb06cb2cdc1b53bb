"""Measurement and diagnostic scripts (see README.md in this directory); nothing here is imported by the product."""

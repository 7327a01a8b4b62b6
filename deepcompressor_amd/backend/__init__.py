"""Packers that turn the calibrated checkpoints into the storage formats of serving back ends."""

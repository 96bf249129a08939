"""Exception types with the reference's names (toppra/exceptions.py:4-13)."""


class ToppraError(Exception):
    """Generic error of the TOPP-RA path."""


class BadInputVelocities(ToppraError):
    """Negative boundary path velocities were given."""


class SolverNotFound(ToppraError):
    """The requested solver wrapper does not exist."""

"""`genjax.inference.requests` (reference: src/genjax/inference/requests.py): edit requests for inference moves.
`Rejuvenate` — a Metropolis-Hastings move with a custom proposal whose acceptance ratio is returned as a weight
(inference/requests/rejuvenate.py:45-94).  `HMC` is out of scope."""
from .._amd.edit import Rejuvenate

__all__ = ["Rejuvenate"]

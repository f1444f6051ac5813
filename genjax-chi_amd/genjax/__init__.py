"""MI355X-native backend for the GenJAX SMC / ImportanceK hot path (public names mirror
genjax-dev/genjax-chi's `genjax` package for that path)."""

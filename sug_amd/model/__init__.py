"""Host-side mirror of the reference's model/ package for the hot path (same module,
class, function and parameter names), implemented over sug_amd.ops."""

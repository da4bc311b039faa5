"""PointTransformerV3: only the serialization stage is built so far."""

"""Post-training quantisation without TensorFlow (reference: birdnet_stm32/conversion/)."""

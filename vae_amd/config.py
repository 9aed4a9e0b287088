"""Default configuration values of the reference (src/config.py:42-88), as plain dicts with the
same names.  Paths to the authors' data are not reproduced; `train.main` takes the count matrix."""
cell_properties = {
    "Dementia": {"discrete": True, "values": [0, 1], "stop_grad": False},
    "CERAD": {"discrete": False, "values": [-1], "stop_grad": True},
    "BRAAK_AD": {"discrete": False, "values": [-1], "stop_grad": False},
    "Sex": {"discrete": True, "values": ["Male", "Female"], "stop_grad": True},
}
batch_properties = None

dataset_cfg = {"cell_properties": cell_properties, "batch_size": 256, "num_workers": 8,
               "batch_properties": batch_properties, "mixup": False, "group_balancing": "bd"}
model_cfg = {"n_layers": 2, "n_hidden": 512, "n_hidden_decoder": 512, "n_hidden_library": 128, "n_latent": 32,
             "n_latent_cell_decoder": 32, "dropout_rate": 0.5, "input_dropout_rate": 0.5,
             "grad_reverse_dict": {"Sex": 0.2}, "cell_decoder_hidden_layer": False}
task_cfg = {"learning_rate": 5e-4, "warmup_steps": 1000.0, "weight_decay": 0.0, "l1_lambda": 0.0,
            "gene_loss_coeff": 5e-3, "balance_classes": False, "n_epochs_kl_warmup": None,
            "batch_properties": batch_properties, "save_gene_vals": True, "use_gdro": False}
trainer_cfg = {"n_devices": 1, "grad_clip_value": 0.5, "accumulate_grad_batches": 1, "precision": "bf16-mixed"}

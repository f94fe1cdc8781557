"""data/data_switch.py of the reference: data identifier -> data class ('lidc'), plus 'synthetic' (LIDC-shaped generator,
data/synthetic.py) for runs without the data set."""


def data_switch(data_identifier):
    if data_identifier == 'lidc':
        from phiseg_code_amd.data.lidc_data import lidc_data
        return lidc_data
    if data_identifier == 'synthetic':
        from phiseg_code_amd.data.synthetic import SyntheticLIDC
        return SyntheticLIDC
    raise ValueError('Unknown data identifier: %s' % data_identifier)
